#!/usr/bin/env python3
"""(gridcell, time) data -- the reference's (lat, lon, time) fixture layout -- at c2-real size: time of one apply against
the (time, gridcell) layout of the same field.  Run on the GPU box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_toolbox_amd import synth
from climate_toolbox_amd.engine import SparsePlan, synth_field

lat, lon, df = synth.realistic_segments(string_labels=False)
res = {}
for wname, dt in (("areawt", "float32"), ("popwt", "float64")):
    cell, code, w, uniq = synth.code_segments(df, lat, lon, wname, "hierid")
    G, R, T = len(lat) * len(lon), len(uniq), 365
    plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
    X = synth_field(T, G, 7, 288.0, 30.0, dtype=dt)
    XT = X.t().contiguous()
    for name, x, lay in (("TG", X, "TG"), ("GT", XT, "GT")):
        out = torch.empty((T, R), dtype=X.dtype, device="cuda")
        for _ in range(3):
            plan.apply(x, layout=lay, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            plan.apply(x, layout=lay, out=out)
        torch.cuda.synchronize()
        res["%s_%s_ms" % (dt, name)] = (time.perf_counter() - t0) / 20 * 1e3
    plan.close()
print(json.dumps(res))
