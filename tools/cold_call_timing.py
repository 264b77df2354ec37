#!/usr/bin/env python3
"""The FIRST reference-named call on a new weights table (c2-real: 402,586 rows, string labels): where its time goes.
Everything the call does before the kernels -- backup fill, label coding, the cell join, the plan build -- happens once per
(table, aggwt, agglev); later calls find the plan by the table's content.  cProfile top entries + wall clock, device-resident field."""
import cProfile, io, json, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from climate_toolbox_amd import synth, minixr, clear_caches, weighted_aggregate_grid_to_regions, engine

lat, lon, df = synth.realistic_segments(string_labels=True)
G = len(lat) * len(lon)
X = engine.synth_field(365, G, seed=3, base=280.0, amp=60.0, dtype="float32")
ds = minixr.Dataset({"tas": (("time", "lat", "lon"), X.view(365, len(lat), len(lon)))}, coords={"time": np.arange(365), "lat": lat, "lon": lon})
weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "ISO", df)          # (code objects, library, first launches)
torch.cuda.synchronize()
res = {}
for aggwt, agglev in (("areawt", "hierid"), ("popwt", "hierid")):
    clear_caches()
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    out = weighted_aggregate_grid_to_regions(ds, "tas", aggwt, agglev, df)
    pr.disable()
    torch.cuda.synchronize()
    cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = weighted_aggregate_grid_to_regions(ds, "tas", aggwt, agglev, df)
    torch.cuda.synchronize()
    warm = time.perf_counter() - t0
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
    res["%s/%s" % (aggwt, agglev)] = {"cold_call_ms": round(cold * 1e3, 1), "next_call_ms": round(warm * 1e3, 2)}
    sys.stderr.write("==== %s %s: cold %.1f ms, next %.2f ms\n%s\n" % (aggwt, agglev, cold * 1e3, warm * 1e3, s.getvalue()[:5000]))
print(json.dumps(res, indent=1))
