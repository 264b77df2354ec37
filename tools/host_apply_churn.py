#!/usr/bin/env python3
"""Host-resident applies, many in a row: per-call wall time with the scratch pool (csrc/wagg_scratch.hip) and with the pool
emptied after every call -- which is what the library did before it had one (four hipMalloc / hipFree of 0.6 GB and three
stream create / destroy per call).  The driver reclaims freed VRAM lazily: without the pool every ~170th call waits ~2 s in
hipMalloc (tools/diag/malloc_stall.cpp shows the same with nothing but hipMalloc / hipFree).

usage: host_apply_churn.py [calls]      (default 400; c2-real table, X = 365 x 1,036,800 fp32, page-locked in place by the call)
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from climate_toolbox_amd import synth, _lib
from climate_toolbox_amd.engine import SparsePlan

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R = len(lat) * len(lon), len(uniq)
plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
Xh = np.random.default_rng(0).normal(280.0, 20.0, (365, G)).astype(np.float32)      # (the library page-locks it in place: HOST_PIN)
L = _lib.load()


def run(release_each_call):
    for _ in range(5):
        plan.apply_host(Xh, flags=_lib.HOST_PIN)
    ms = []
    for _ in range(calls):
        t0 = time.perf_counter()
        plan.apply_host(Xh, flags=_lib.HOST_PIN)
        ms.append((time.perf_counter() - t0) * 1e3)
        if release_each_call:
            L.wagg_release_scratch()
    a = np.sort(np.asarray(ms))
    return {"calls": calls, "median_ms": round(float(np.median(a)), 3), "p99_ms": round(float(a[int(0.99 * (len(a) - 1))]), 3),
            "max_ms": round(float(a[-1]), 3), "calls_over_100_ms": int((a > 100.0).sum()), "total_s": round(float(a.sum()) * 1e-3, 3),
            "h2d_gbs_at_median": round(Xh.nbytes / (float(np.median(a)) * 1e-3) * 1e-9, 1)}


res = {"pool": run(False), "pool_emptied_after_every_call": run(True),
       "what": "SparsePlan.apply_host on the c2-real table, X 1.514 GB (WAGG_HOST_PIN), result 35.6 MB; wall ms per call"}
print(json.dumps(res, indent=1))
