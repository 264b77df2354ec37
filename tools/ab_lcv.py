#!/usr/bin/env python3
"""A/B of the segment-table kernels between two builds of the library on ONE box: `python tools/ab_lcv.py <lib file name>`
(a file in climate_toolbox_amd/lib/) times the c2-real / c3-real device applies -- plain fp32 and fp64, fused powers 1..4,
degree days with one and three thresholds -- as medians of 300 launches after 0.3 s of warm-up, and prints one JSON object.
Run the two libraries alternately (A B A B) in fresh processes and compare.  `--nan-ocean`: every cell the table does not
reference is NaN (a land-only dataset): before round 6 a NaN anywhere in an item's lines sent the item down the general
(NaN-testing) forms; since the unreferenced-quad flags only referenced data decides."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from climate_toolbox_amd import _lib
NAN_OCEAN = "--nan-ocean" in sys.argv
if NAN_OCEAN:
    sys.argv.remove("--nan-ocean")
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.path.basename(sys.argv[1]))
from climate_toolbox_amd import engine, synth

lat, lon, df = synth.realistic_segments(string_labels=False)
G, T = len(lat) * len(lon), 365
res = {"lib": os.path.basename(_lib.LIB_PATH), "nan_ocean": NAN_OCEAN}


def med(fn, n=300):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        fn()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ms[len(ms) // 2], 4)


for dtype, wname in (("float32", "areawt"), ("float64", "popwt")):
    cell, code, w, uniq = synth.code_segments(df, lat, lon, wname, "hierid")
    R = len(uniq)
    plan = engine.SparsePlan(cell, code, w, G, R, row_len=len(lon))
    X = engine.synth_field(T, G, seed=1000, base=280.0, amp=60.0, dtype=dtype)
    Xh = X + engine.synth_field(T, G, seed=2000, base=6.0, amp=10.0, dtype=dtype)
    if NAN_OCEAN:
        ocean = torch.ones(G, dtype=torch.bool, device="cuda")
        ocean[torch.from_numpy(np.unique(cell)).cuda().long()] = False
        X[:, ocean] = float("nan")
        Xh[:, ocean] = float("nan")
    o1 = torch.empty((T, R), dtype=X.dtype, device="cuda")
    o4 = torch.empty((4, T, R), dtype=X.dtype, device="cuda")
    o3 = torch.empty((3, T, R), dtype=X.dtype, device="cuda")
    k = "f32" if dtype == "float32" else "f64"
    res["plain_" + k] = med(lambda: plan.apply(X, out=o1))
    res["poly4_" + k] = med(lambda: plan.apply_poly(X, -273.15, 4, out=o4))
    res["edd1_" + k] = med(lambda: plan.apply_edd(X, Xh, [10.0], offset=-273.15, out=o3[:1]))
    res["edd3_" + k] = med(lambda: plan.apply_edd(X, Xh, [10.0, 20.0, 30.0], offset=-273.15, out=o3))
    XT = X.t().contiguous()
    res["plain_GT_" + k] = med(lambda: plan.apply(XT, layout="GT", out=o1))
    del X, Xh, XT
    plan.close()
print(json.dumps(res))
