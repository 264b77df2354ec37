#!/usr/bin/env python3
"""Times the fused tas_poly aggregation (wagg_apply_poly_f32, powers 1..K in one pass over X)
against K separate aggregations of pre-transformed grids, on c2-real.  Run on the GPU box."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from climate_toolbox_amd import synth
from climate_toolbox_amd.engine import SparsePlan, synth_field

lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R, T = len(lat) * len(lon), len(uniq), 365
plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
X = synth_field(T, G, 7, 288.0, 30.0)
res = {}
for K in (1, 2, 3, 4, 5):
    out = torch.empty((K, T, R), dtype=torch.float32, device="cuda")
    for _ in range(3):
        plan.apply_poly(X, -273.15, K, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        plan.apply_poly(X, -273.15, K, out=out)
    torch.cuda.synchronize(); res["fused_K%d_ms" % K] = (time.perf_counter() - t0) / 20 * 1e3
# the unfused pipeline: transform the grid (one elementwise pass per power), then aggregate
K = 4
out = torch.empty((T, R), dtype=torch.float32, device="cuda")
def unfused():
    y = X - 273.15
    for p in range(1, K + 1):
        plan.apply(y ** p if p > 1 else y, out=out)
for _ in range(2): unfused()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): unfused()
torch.cuda.synchronize(); res["unfused_K4_ms"] = (time.perf_counter() - t0) / 10 * 1e3
print(json.dumps(res))
