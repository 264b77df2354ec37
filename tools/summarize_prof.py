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
    for (k, c), v in agg.items():
        if kernel_sub in k and c == counter:
            return sum(v) / len(v)
    return None


tj = os.path.join(dst, "traffic.json")
traffic = json.load(open(tj)) if os.path.exists(tj) else {}
WIDE = ("FETCH_SIZE x2 (gfx950 tallies the 128-B requests of a 16-B/lane contiguous stream at 64 B; for the dense kernel "
        "the raw value is below the bytes of W that must be read), WRITE_SIZE exact")
GATHER = ("FETCH_SIZE x1: calibrated on the segment-table gather in round 1 with every grid cell referenced (land_frac=1.0): "
          "raw 1.567 GB vs 1.514 GB of X, TCC_EA0_RDREQ x 64 B = 1.567 GB; WRITE_SIZE exact")
for wl, ksub, wide in (("c2-dense", "dense_mfma_kernel<float", True), ("c2-real", "sparse_lc_kernel<true, 1, false>", False),
                       ("c3-real", "sparse_stream_kernel<double", False), ("c5-uniform", "spmm_kernel", True)):
    fs, ws = mean(ksub, "FETCH_SIZE"), mean(ksub, "WRITE_SIZE")
    if fs is None or ws is None:
        continue
    traffic[wl] = {
        "hbm_bytes_per_launch": (2.0 if wide else 1.0) * fs * 1024 + ws * 1024,
        "fetch_size_kib_raw": fs, "write_size_kib_raw": ws,
        "correction": (WIDE if wide else GATHER) if wl != "c5-uniform" else
                      ("FETCH_SIZE x2 applied to the whole figure: exact for the X tiles (16-B/lane LDS-DMA), an upper bound for "
                       "the entry lists (4-B/lane loads, uncalibrated); the bytes that must move are ~19 GB of X + 36 GB of lists"),
        "source": "profiles/%s_pmc.csv" % tag, "measured": "round %d" % int(tag.lstrip("r")),
    }
json.dump(traffic, open(tj, "w"), indent=1)
print(open(os.path.join(dst, tag + "_kernel_stats.csv")).read()[:1500])
print(json.dumps(traffic, indent=1))
