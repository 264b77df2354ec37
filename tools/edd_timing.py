#!/usr/bin/env python3
"""Times the fused Snyder degree-day aggregation (wagg_apply_edd_f32) against transforming the
grids on the device first and aggregating the result, on c2-real.  Run on the GPU box."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from climate_toolbox_amd import engine, synth
from climate_toolbox_amd.engine import SparsePlan, synth_field

lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R, T = len(lat) * len(lon), len(uniq), 365
plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
tmin = synth_field(T, G, 7, 288.0, 30.0)
tmax = tmin + synth_field(T, G, 8, 6.0, 10.0)
res = {}
for K in (1, 3):
    thr = [10.0, 20.0, 30.0][:K]
    out = torch.empty((K, T, R), dtype=torch.float32, device="cuda")
    for _ in range(3): plan.apply_edd(tmin, tmax, thr, offset=-273.15, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): plan.apply_edd(tmin, tmax, thr, offset=-273.15, out=out)
    torch.cuda.synchronize(); res["fused_K%d_ms" % K] = (time.perf_counter() - t0) / 10 * 1e3
o1 = torch.empty((T, R), dtype=torch.float32, device="cuda")
def unfused():
    a, b = tmin - 273.15, tmax - 273.15
    plan.apply(engine.transform_edd(a, b, 0.0, [(1.0, 30.0)]), out=o1)
for _ in range(2): unfused()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): unfused()
torch.cuda.synchronize(); res["unfused_K1_ms"] = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps(res))
