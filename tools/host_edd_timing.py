#!/usr/bin/env python3
"""The reference-named calls on HOST-resident c2-real fields that no row-block pipeline serves -- Snyder degree days (two
fields), the (lat, lon, time) layout of the reference's fixture -- end to end through the drop-in: what the upload costs now
(wagg_upload: page-locked in place, one DMA) against a pageable runtime copy of the same array.  Run on the GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_toolbox_amd import engine, synth

G = 720 * 1440
X = (280 + 20 * np.random.default_rng(0).standard_normal((365, G), dtype=np.float32))
res = {}


def med(fn, reps=5):
    fn(); fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(1e3 * sorted(ts)[len(ts) // 2], 2)


res["upload_1.5GB_ms"] = med(lambda: engine.upload(X))
res["pageable_runtime_copy_1.5GB_ms"] = med(lambda: torch.from_numpy(X).cuda())
print(json.dumps(res))
