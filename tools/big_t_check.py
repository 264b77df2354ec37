#!/usr/bin/env python3
"""The c4 row count (T = 10,950) through the segment-table form of the c2-real table on one GPU: time of one apply, the
same rows in two halves bit-equal, fp64 on a 2,000-row slice consistent.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from climate_toolbox_amd import synth
from climate_toolbox_amd.engine import SparsePlan, synth_field
lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R, T = len(lat) * len(lon), len(uniq), 10950
plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
X = synth_field(T, G, 3, 288.0, 30.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
full = plan.apply(X)
torch.cuda.synchronize(); t1 = time.perf_counter()
full = plan.apply(X)
torch.cuda.synchronize(); t2 = time.perf_counter()
a = plan.apply(X[:5000]); b = plan.apply(X[5000:])
assert torch.equal(full[:5000], a) and torch.equal(full[5000:], b)
print("T=10950 apply %.2f ms (first %.2f), halves bit-equal" % ((t2 - t1) * 1e3, (t1 - t0) * 1e3))
Xd = X[:2000].double()
fd = plan.apply(Xd)
assert torch.allclose(fd.float(), full[:2000], rtol=2e-5, atol=1e-3)
print("fp64 T=2000 ok")
