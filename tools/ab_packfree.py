#!/usr/bin/env python3
"""A/B on one GPU: full-form dense apply with the pack-free first pass vs the packed pass (diagnostic
library, WAGG_DENSE_PACKED=1), for several row counts.  usage: python tools/ab_packfree.py [R]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_toolbox_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libwagg_diag.so")
from climate_toolbox_amd import engine

G, R = 720 * 1440, int(sys.argv[1]) if len(sys.argv) > 1 else 24378
plan = engine.DensePlan.synth(G, R, seed=1)
for T in (365, 704, 1056, 1369):
    X = engine.synth_field(T, G, seed=3, base=280.0, amp=20.0, dtype="float32")
    for mode in ("rm", "packed", "rm", "packed"):
        if mode == "packed":
            os.environ["WAGG_DENSE_PACKED"] = "1"
        else:
            os.environ.pop("WAGG_DENSE_PACKED", None)
        plan.apply(X)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            plan.apply(X)
        torch.cuda.synchronize()
        print("T=%d %s %.2f ms" % (T, mode, (time.perf_counter() - t0) / 2 * 1e3), flush=True)
    del X
