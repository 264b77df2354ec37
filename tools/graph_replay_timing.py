#!/usr/bin/env python3
"""c1 (the reference's CPU-sized case: 365 x 16,200 cells, 100 regions, fp64) is launch-bound on the GPU: step time
of eager applies vs one captured HIP graph replayed (torch.cuda.CUDAGraph on a side stream).  Run on the GPU box."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from climate_toolbox_amd import synth
from climate_toolbox_amd.engine import SparsePlan

lat, lon, tas, df = synth.c1_workload(T=365)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R, T = len(lat) * len(lon), len(uniq), 365
plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
X = torch.from_numpy(np.ascontiguousarray(tas.reshape(T, G))).cuda()
out = torch.empty((T, R), dtype=X.dtype, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        plan.apply(X, out=out)
s.synchronize()
N = 200
res = {}
with torch.cuda.stream(s):
    t0 = time.perf_counter()
    for _ in range(N):
        plan.apply(X, out=out)
    s.synchronize()
    res["eager_ms_per_apply"] = (time.perf_counter() - t0) / N * 1e3
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    for _ in range(10):
        plan.apply(X, out=out)
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N // 10):
    g.replay()
torch.cuda.synchronize()
res["graph_ms_per_apply"] = (time.perf_counter() - t0) / N * 1e3
print(json.dumps(res))
