#!/usr/bin/env python3
"""Country-level aggregation (agglev = ISO) of c2-real through the segment-table plan (giant groups) and through the
tile-sparse dense form built from the same table; plus what the drop-in picks.  GPU box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from climate_toolbox_amd import synth, aggregations as A
from climate_toolbox_amd.engine import DensePlan, SparsePlan, synth_field

lat, lon, df = synth.realistic_segments(string_labels=False)
G, T = len(lat) * len(lon), 365
X = synth_field(T, G, 7, 288.0, 30.0)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "ISO")
R = len(uniq)
ref = None
res = {"R": R}
for name, mk in (("segment_table", lambda: SparsePlan(cell, code, w, G, R, row_len=len(lon))),
                 ("dense_from_segments", lambda: DensePlan.from_segments(cell, code, w, G, R))):
    plan = mk()
    out = torch.empty((T, R), dtype=torch.float32, device="cuda")
    for _ in range(3): plan.apply(X, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): plan.apply(X, out=out)
    torch.cuda.synchronize()
    got = out[:8].cpu().numpy()
    if ref is None:
        ref = got                                  # the two forms against each other (parity vs the oracle: tests/)
    res[name] = {"ms": (time.perf_counter() - t0) / 20 * 1e3, "info": {k: int(v) for k, v in plan.info.items()},
                 "max_rel_diff_vs_segment_table": float(np.nanmax(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)))}
free, total = torch.cuda.mem_get_info()
sp = SparsePlan(cell, code, w, G, R, row_len=len(lon))
res["dropin_prefers_dense"] = bool(A._prefer_dense(sp.info["n_ucells"], G, R, True, "TG", free))
res["n_ucells_over_G"] = sp.info["n_ucells"] / G
print(json.dumps(res))
