#!/usr/bin/env python3
"""Where a c2-real step's time goes besides the kernel: wall time per apply for (T,R) and (R,T) output,
with and without a device sync per call (GPU box)."""
import json, sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from climate_toolbox_amd import engine, synth
lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R, T = len(lat) * len(lon), len(uniq), 365
plan = engine.SparsePlan(cell, code, w, G, R, row_len=len(lon))
res = {}
for dt in ("float32", "float64"):
    X = engine.synth_field(T, G, 1, 280.0, 60.0, dtype=dt)
    for lay in ("TR", "RT"):
        out = torch.empty((T, R) if lay == "TR" else (R, T), dtype=X.dtype, device="cuda")
        for _ in range(5): plan.apply(X, out=out, out_layout=lay)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n): plan.apply(X, out=out, out_layout=lay)
        t_enq = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / n
        res["%s_%s" % (dt, lay)] = {"enqueue_ms": round(t_enq * 1e3, 4), "wall_ms": round(t_all * 1e3, 4)}
print(json.dumps(res))
