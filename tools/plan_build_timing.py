#!/usr/bin/env python3
"""Time of wagg_plan_create on the c2-real table (400k segments, 24,378 regions) with and without the whole-line
chunkings.  Run on the GPU box."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from climate_toolbox_amd import synth, _lib
from climate_toolbox_amd.engine import SparsePlan
lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R = len(lat) * len(lon), len(uniq)
torch.zeros(1, device="cuda")
for flags in (0, _lib.PLAN_NO_LINES, 0, _lib.PLAN_NO_LINES):
    t0 = time.perf_counter()
    p = SparsePlan(cell, code, w, G, R, row_len=len(lon), flags=flags)
    torch.cuda.synchronize()
    print("flags", flags, "plan create %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    p.close()
