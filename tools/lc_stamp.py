#!/usr/bin/env python3
"""Phase stamps of the loader/consumer kernel on c2-real (diagnostic library, WAGG_SPARSE_STAMP=1), optionally with
parts of the kernel left out or changed (WAGG_LC_KNOB: 128 = no result stores, 256 = no segment walk, 1 = consumers idle,
2 = non-temporal loads (results stay right), 16384 = rows sent straight to the image by LDS-DMA; results are wrong with
all but 2 -- timing only; WAGG_LCV_NW = number of workgroups, i.e. CUs used).  Run on the GPU box after `make -C climate_toolbox_amd/csrc diag`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_toolbox_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libwagg_diag.so")
from climate_toolbox_amd import synth
from climate_toolbox_amd.engine import SparsePlan, synth_field

lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R, T = len(lat) * len(lon), len(uniq), 365
flags = int(os.environ.get("PLAN_FLAGS", "0"))
plan = SparsePlan(cell, code, w, G, R, row_len=len(lon), flags=flags)
X = synth_field(T, G, 7, 288.0, 30.0)
o = torch.empty((T, R), dtype=torch.float32, device="cuda")
for knob in [int(k) for k in os.environ.get("KNOBS", "0,128,256,384,1").split(",")]:
    os.environ["WAGG_LC_KNOB"] = str(knob)
    os.environ.pop("WAGG_SPARSE_STAMP", None)
    for _ in range(3):
        plan.apply(X, out=o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.apply(X, out=o)
    torch.cuda.synchronize()
    print("flags=%d knob=%d step %.4f ms" % (flags, knob, (time.perf_counter() - t0) / 20 * 1e3), file=sys.stderr, flush=True)
    os.environ["WAGG_SPARSE_STAMP"] = "1"
    plan.apply(X, out=o)
    torch.cuda.synchronize()
