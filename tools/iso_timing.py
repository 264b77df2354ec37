#!/usr/bin/env python3
"""c2-real aggregated to the COUNTRY level (agglev = ISO: ~200 regions of thousands of cells each,
every one a multi-chunk "giant" group) next to the impact-region level (hierid).  GPU box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from climate_toolbox_amd import synth
from climate_toolbox_amd.engine import SparsePlan, synth_field

lat, lon, df = synth.realistic_segments(string_labels=False)
G, T = len(lat) * len(lon), 365
X = synth_field(T, G, 7, 288.0, 30.0)
res = {}
for lev in ("hierid", "ISO"):
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", lev)
    plan = SparsePlan(cell, code, w, G, len(uniq), row_len=len(lon))
    out = torch.empty((T, len(uniq)), dtype=torch.float32, device="cuda")
    for _ in range(3): plan.apply(X, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): plan.apply(X, out=out)
    torch.cuda.synchronize()
    res[lev] = {"R": len(uniq), "ms": (time.perf_counter() - t0) / 20 * 1e3, "n_giant": plan.info["n_giant"],
                "n_chunks": plan.info["n_chunks"], "n_groups": plan.info["n_groups"]}
print(json.dumps(res))
