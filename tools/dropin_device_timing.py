"""End-to-end time of repeated calls of the reference-named function on a DEVICE-resident field (c2-real, cached plan,
per-table memo) and a cProfile of where the host time goes.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from climate_toolbox_amd import engine, minixr, synth, weighted_aggregate_grid_to_regions
lat, lon, df = synth.realistic_segments()
T = 365
Xd = engine.synth_field(T, len(lat) * len(lon), seed=3, base=280.0, amp=60.0).reshape(T, len(lat), len(lon))
ds = minixr.Dataset({"tas": (("time", "lat", "lon"), Xd)}, coords={"lat": lat, "lon": lon})
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
    torch.cuda.synchronize()
    print("device-resident call %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5): out = weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
