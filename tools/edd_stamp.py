#!/usr/bin/env python3
"""Phase stamps of the fused degree-day kernel (diagnostic library, WAGG_SPARSE_STAMP=1) on c2-real: where a
loader wave's cycles go per stage (issue / wait + arithmetic + park / barrier) next to the consumers'.
Run on the GPU box after `make -C climate_toolbox_amd/csrc diag`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_toolbox_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libwagg_diag.so")
from climate_toolbox_amd import synth
from climate_toolbox_amd.engine import SparsePlan, synth_field

lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R, T = len(lat) * len(lon), len(uniq), 365
plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
tmin = synth_field(T, G, 7, 288.0, 30.0)
tmax = tmin + synth_field(T, G, 8, 6.0, 10.0)
for K in (1, 3):
    out = torch.empty((K, T, R), dtype=torch.float32, device="cuda")
    thr = [10.0, 20.0, 30.0][:K]
    plan.apply_edd(tmin, tmax, thr, offset=-273.15, out=out)
    torch.cuda.synchronize()
    os.environ["WAGG_SPARSE_STAMP"] = "1"
    print("K=%d" % K, file=sys.stderr, flush=True)
    plan.apply_edd(tmin, tmax, thr, offset=-273.15, out=out)
    torch.cuda.synchronize()
    os.environ.pop("WAGG_SPARSE_STAMP")
o = torch.empty((T, R), dtype=torch.float32, device="cuda")
plan.apply(tmin, out=o)
torch.cuda.synchronize()
os.environ["WAGG_SPARSE_STAMP"] = "1"
print("plain", file=sys.stderr, flush=True)
plan.apply(tmin, out=o)
torch.cuda.synchronize()
