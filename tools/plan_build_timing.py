#!/usr/bin/env python3
"""Plan-build times on the GPU box.

  (1) wagg_plan_create on the c2-real table (400k segments, 24,378 regions) with and without the whole-line chunkings,
      threaded and on the calling thread alone (WAGG_PLAN_SERIAL_BUILD).
  (2) wagg_dense_create_from_csr on the c5 tables (2.5e8 entries, 3 GB of host arrays): wall clock, the library's own split.
  (3) VERDICT r4 item 6: the c5-uniform build WHILE another thread applies the c2-real plan on a stream of its own --
      the applies' kernel times (event pairs) with and without the build beside them, and the build's time with and
      without the applies.  A build runs on its own non-blocking stream out of one arena and never synchronises the device
      while it works, so the applies keep flowing; what they lose is what two jobs sharing one GPU's HBM lose.
"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from climate_toolbox_amd import synth, _lib, engine
from climate_toolbox_amd.engine import DensePlan, SparsePlan

lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R = len(lat) * len(lon), len(uniq)
torch.zeros(1, device="cuda")
res = {"plan_create_ms": {}}
for name, flags in (("lines, threaded", 0), ("no lines", _lib.PLAN_NO_LINES), ("lines, serial", _lib.PLAN_SERIAL_BUILD), ("lines, threaded (2)", 0)):
    t0 = time.perf_counter()
    p = SparsePlan(cell, code, w, G, R, row_len=len(lon), flags=flags)
    torch.cuda.synchronize()
    res["plan_create_ms"][name] = round((time.perf_counter() - t0) * 1e3, 1)
    p.close()

tables = {}
for bl, fill in ((False, 0.01), (True, 0.952)):
    tables[bl] = engine.synth_table_csr(G, R, 2, fill, blocklocal=bl)
res["from_csr_s"] = {}
for bl in (False, True):
    for dt in ("float32", "float64"):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            p = DensePlan.from_csr(*tables[bl], G, R, dtype=dt)
            ts.append((time.perf_counter() - t0, p.info["build_s"], p.info["build_upload_s"], p.info["form"]))
            p.close()
        best = min(ts)
        res["from_csr_s"]["%s %s" % ("block-local" if bl else "uniform", dt)] = {
            "wall_s": [round(t[0], 4) for t in ts], "library_s": round(best[1], 4), "upload_s": round(best[2], 4), "form": best[3]}

# (3) a build beside applies on another stream
sp = SparsePlan(cell, code, w, G, R, row_len=len(lon))
X = engine.synth_field(365, G, seed=3, base=280.0, amp=60.0, dtype="float32")
out = torch.empty((365, R), dtype=torch.float32, device="cuda")


def apply_times(seconds, stop=None):
    """kernel + combine of c2-real applies on a stream of this thread, one event pair per apply, for `seconds` (or until stop)"""
    s = torch.cuda.Stream()
    ms = []
    t_end = time.perf_counter() + seconds
    with torch.cuda.stream(s):
        for _ in range(50):
            sp.apply(X, out=out, stream=s)
        s.synchronize()
        while time.perf_counter() < t_end and not (stop is not None and stop.is_set()):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for a, b in ev:
                a.record(s); sp.apply(X, out=out, stream=s); b.record(s)
            s.synchronize()
            ms += [a.elapsed_time(b) for a, b in ev]
    return ms


def pct(v, q):
    v = sorted(v)
    return round(v[min(len(v) - 1, int(q * len(v)))], 4)


alone = apply_times(1.0)
box = {}
th = threading.Thread(target=lambda: box.setdefault("ms", apply_times(30.0, stop)))
stop = threading.Event()
th.start()
time.sleep(0.3)
builds = []
for _ in range(3):
    t0 = time.perf_counter()
    p = DensePlan.from_csr(*tables[False], G, R)
    builds.append(round(time.perf_counter() - t0, 4))
    p.close()
    time.sleep(0.1)
stop.set()
th.join()
beside = box["ms"]
res["build_beside_applies"] = {
    "apply_ms_alone": {"n": len(alone), "p50": pct(alone, 0.5), "p90": pct(alone, 0.9), "p99": pct(alone, 0.99), "max": round(max(alone), 4)},
    "apply_ms_beside_3_builds": {"n": len(beside), "p50": pct(beside, 0.5), "p90": pct(beside, 0.9), "p99": pct(beside, 0.99),
                                 "max": round(max(beside), 4)},
    "c5_uniform_build_s_beside_applies": builds, "c5_uniform_build_s_alone": res["from_csr_s"]["uniform float32"]["wall_s"],
    "what": "c2-real applies (kernel + combine, event pairs on their own stream) while another thread builds the c5-uniform plan "
            "three times; an apply that overlaps a build shares the GPU with its sort passes"}
print(json.dumps(res, indent=1))
