#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (tools/gpu_profile.sh) into profiles/<tag>_*.

usage: python tools/summarize_prof.py r02
Writes profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 --stats table of the default bench run),
profiles/<tag>_<workload>_kernel_stats.csv for every trace of tools/gpu_profile_extra.sh,
profiles/<tag>_pmc.csv (per kernel and counter: launches, mean value) and updates
profiles/traffic.json (HBM bytes per launch of the dominant kernels, corrected as
MI355X_MICROARCH.md prescribes: FETCH_SIZE is in KiB and reads 1/2 of a wide (16 B/lane) coalesced
stream on gfx950, WRITE_SIZE is exact), each entry stamped with the round it was measured in.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "trace_bench.json")):
    shutil.copy(os.path.join(src, "trace_bench.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
extra = os.path.join(ROOT, "gpurun_out", "prof_extra_" + tag)
for f in sorted(glob.glob(os.path.join(extra, "*", "t_kernel_stats.csv"))):
    name = os.path.basename(os.path.dirname(f))
    shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
    j = os.path.join(extra, name + ".json")
    if os.path.exists(j) and os.path.getsize(j):
        shutil.copy(j, os.path.join(dst, "%s_%s_under_rocprof.json" % (tag, name)))
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc_*", "bench_counter_collection.csv")) + \
        glob.glob(os.path.join(extra, "pmc_*", "bench_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(os.path.join(dst, tag + "_pmc.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "mean"])
    for (k, c), v in sorted(agg.items()):
        w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])


def mean(kernel_sub, counter):
    """mean counter value per launch of the kernel.  c1 and c3-real run the SAME instantiation
    (sparse_lcv_kernel<double, true, 1, false, false>): "<name>#big" / "<name>#small" select the launches above / below half of the
    largest value (c3: 1M cells, c1: 16,200)."""
    pick = None
    if "#" in kernel_sub:
        kernel_sub, pick = kernel_sub.split("#")
    for (k, c), v in agg.items():
        if kernel_sub in k and c == counter:
            if pick:
                top = max(v)
                v = [x for x in v if (x >= 0.5 * top) == (pick == "big")]
            return sum(v) / len(v) if v else None
    return None


# per workload: the dominant kernel's launches and mean duration from the kernel trace (the --stats table averages c1
# and c3-real together: one kernel instantiation serves both)
tr = os.path.join(src, "trace", "bench_kernel_trace.csv")
if os.path.exists(tr):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(tr)):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    rows = []
    for wl, ksub in (("c2-dense", "dense_mfma_kernel<float, 0, false, 23"), ("c4 (rank share): 3 full row blocks", "dense_mfma_kernel<float, 0, false, 22, true"),
                     ("c4 (rank share): the remainder (313 rows)", "dense_mfma_kernel<float, 0, false, 20, false"),
                     # (round 4: the default line's host-resident / drop-in / country-level legs launch the c2-real instantiation on row blocks
                     #  and on another table as well: "#big" keeps the full-field launches of the c2-real table)
                     ("c2-real", "sparse_lcv_kernel<float, true, 1, false, false>#big"), ("c3-real", "sparse_lcv_kernel<double, true, 1, false, false>#big"),
                     ("c1", "sparse_lcv_kernel<double, true, 1, false, false>#small"), ("c5-block: 6 full row blocks", "dense_pieces_kernel<float, 21"), ("c5-block: the remainder (266 rows)", "dense_pieces_kernel<float, 18"),
                     ("c5-block-f64", "dense_pieces_kernel<double"), ("c5-uniform", "spmm_kernel<float>"),
                     ("c5-uniform-f64", "spmm_kernel<double>"), # (round 5, third part: the host forms of the fused transforms launch the same instantiations on 64-row blocks: "#big")
                     ("c2-real fused tas_poly 1..4", "sparse_lcv_kernel<float, true, 4, false, false>#big"),
                     ("c2-real fused snyder_edd, one threshold", "sparse_lcv_kernel<float, true, 1, true, false>#big"),
                     ("c2-real fused snyder_edd, three thresholds", "sparse_lcv_kernel<float, true, 3, true, false>"),
                     ("c2-real combine + transpose, one plane", "combine_parts_kernel<float, true>#band:0.02:0.04"),
                     ("c2-real combine + transpose, 3-4 planes (fused transforms)", "combine_parts_kernel<float, true>#big"),
                     ("c3-real combine + transpose, one plane", "combine_parts_kernel<double, true>#band:0.035:0.07"),
                     ("c3-real combine + transpose, 3-4 planes (fused transforms)", "combine_parts_kernel<double, true>#big"),
                     # round 4: plans built on the device from the caller's CSR table (c5: 2.5e8 entries; four plans per run)
                     ("c5 table -> plan: sort keys", "keygen_kernel"), ("c5 table -> plan: chunk bounds", "chunk_first_kernel"),
                     ("c5 table -> plan: one-pass sort, bin counts per chunk", "chunk_hist_kernel"),
                     ("c5 table -> plan: one-pass sort, stable partition per chunk", "chunk_scatter_kernel"),
                     ("c5 table -> plan: radix histogram (per pass; the denominators' sort by region, and the general sort)", "rs_hist_kernel"),
                     ("c5 table -> plan: radix scatter (per pass; as above)", "rs_scatter_kernel"), ("c5 table -> plan: coalesce duplicates", "coalesce_kernel"),
                     ("c5 table -> plan: denominators", "den_kernel"), ("c5 table -> plan: tile census", "table_tiles_kernel"),
                     ("c5 table -> plan: list bounds (binary search per bucket; also the form choice's cost census)", "spmm_bounds_kernel"),
                     ("c5 table -> plan: list cost census", "spmm_cost_kernel"),
                     ("c5 table -> plan: entry lists", "spmm_fill_kernel"), ("c5 table -> plan: list padding", "spmm_pad_kernel"),
                     ("c5 table -> plan: packed tiles", "table_scatter_kernel"),
                     ("c5 table -> plan: scan (per call)", "scan_apply_kernel"),
                     ("synthetic c5 table (CSR generator)", "synth_csr_kernel")):
        pick = None
        if "#" in ksub:
            ksub, pick = ksub.split("#")
        for k, v in dur.items():
            if ksub in k:
                if pick and pick.startswith("band:"):       # durations inside [lo, hi] ms
                    lo, hi = (float(x) for x in pick.split(":")[1:])
                    v = [x for x in v if lo <= x <= hi]
                elif pick:
                    top = max(v)
                    v = [x for x in v if (x >= 0.5 * top) == (pick == "big")]
                if v:
                    rows.append((wl, k.replace("void wagg::", ""), len(v), sum(v) / len(v), min(v), max(v)))
    with open(os.path.join(dst, tag + "_kernel_by_workload.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["workload", "dominant kernel", "launches", "avg_ms", "min_ms", "max_ms"])
        for r in rows:
            w.writerow([r[0], r[1], r[2]] + ["%.4f" % x for x in r[3:]])


tj = os.path.join(dst, "traffic.json")
traffic = json.load(open(tj)) if os.path.exists(tj) else {}
WIDE = ("FETCH_SIZE x2 (gfx950 tallies the 128-B requests of a 16-B/lane contiguous stream at 64 B: MI355X_MICROARCH.md, HBM), "
        "WRITE_SIZE exact")
GATHER = ("FETCH_SIZE x1: calibrated on the segment-table gather in round 1 with every grid cell referenced (land_frac=1.0): "
          "raw 1.567 GB vs 1.514 GB of X, TCC_EA0_RDREQ x 64 B = 1.567 GB; WRITE_SIZE exact")
LINES = ("whole-line chunks: every load instruction reads eight whole 128-B lines, i.e. the wide coalesced case: FETCH_SIZE x2 if "
         "the raw figure is about half of the lines' bytes (lines_ucells x 4 B x T for fp32, lines64_ucells x 8 B x T for fp64, in "
         "the bench line's plan), x1 if it matches them: see `calibration`; WRITE_SIZE exact")
for wl, ksub, mode in (("c2-dense", "dense_mfma_kernel<float, 0, false, 23", "wide"), ("c4", "dense_mfma_kernel<float, 0, false, 22, true+dense_mfma_kernel<float, 0, false, 20, false", "wide"),
                       ("c2-real", "sparse_lcv_kernel<float, true, 1, false, false>#big", "lines"), ("c3-real", "sparse_lcv_kernel<double, true, 1, false, false>#big", "lines64"),
                       ("c5-block", "dense_pieces_kernel<float, 21+dense_pieces_kernel<float, 18", "wide"), ("c5-block-f64", "dense_pieces_kernel<double", "wide"),
                       ("c5-uniform", "spmm_kernel<float>", "wide"), ("c5-uniform-f64", "spmm_kernel<double>", "wide")):
    # "a+b": an apply that launches its dominant kernel twice (full row blocks, then the remainder): the sum of the two means
    parts = [(mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")) for k in ksub.split("+")]
    if any(f is None or w_ is None for f, w_ in parts):
        continue
    fs, ws = sum(f for f, _ in parts), sum(w_ for _, w_ in parts)
    factor = 1.0 if mode == "gather" else 2.0      # gather64: see below
    entry = {"fetch_size_kib_raw": fs, "write_size_kib_raw": ws, "source": "profiles/%s_pmc.csv" % tag,
             "measured": "round %d" % int(tag.lstrip("r"))}
    if mode in ("lines", "lines64"):
        # known bytes of the whole-line gather: 20,607 lines x 128 B x 365 rows for c2-real (bench line: plan.lines_ucells),
        # 30,527 x 128 B x 365 for c3-real (plan.lines64_ucells)
        known = float(os.environ.get("LINES_BYTES" if mode == "lines" else "LINES64_BYTES", "0")) or None
        if known:
            factor = 2.0 if fs * 1024 < 0.75 * known else 1.0
            entry["calibration"] = {"lines_bytes": known, "raw_over_lines": fs * 1024 / known, "factor": factor}
    entry["hbm_bytes_per_launch"] = factor * fs * 1024 + ws * 1024
    entry["correction"] = {"wide": WIDE, "gather": GATHER, "lines": LINES, "lines64": LINES}[mode] if not wl.startswith("c5-uniform") else (
        "FETCH_SIZE x2 applied to the whole figure: exact for the X tiles and the weight lists (16-B/lane LDS-DMA), an upper bound for "
        "the lo16 pair lists (4-B/lane loads, uncalibrated)")
    traffic[wl] = entry
json.dump(traffic, open(tj, "w"), indent=1)
print(open(os.path.join(dst, tag + "_kernel_stats.csv")).read()[:1500])
print(json.dumps(traffic, indent=1))
