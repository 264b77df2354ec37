"""What bench.py writes, taken apart (tests only): the compact line on stdout, the detail on stderr."""
import json

COMPACT_LIMIT = 6144          # bytes; bench.py's own limit (the round-5 driver record failed to parse a 28.5 KB line)


def split_bench_output(proc):
    """(compact record, detail record or None) of a finished `bench.py` child: stdout must hold exactly ONE line, shorter
    than COMPACT_LIMIT, that parses; the detail is the `BENCH_DETAIL {...}` line on stderr (absent for --dry-run)."""
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout[-2000:]
    assert len(lines[0]) < COMPACT_LIMIT, len(lines[0])
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert "workload" in rec["config"]
    detail = None
    for ln in proc.stderr.splitlines():
        if ln.startswith("BENCH_DETAIL "):
            detail = json.loads(ln[len("BENCH_DETAIL "):])
    return rec, detail
