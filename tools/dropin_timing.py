"""End-to-end timing of the drop-in function on host data (c2-real, fp32): first call (plan build)
and repeated call (plan cache hit).  usage: python tools/dropin_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from climate_toolbox_amd import engine, minixr, synth, weighted_aggregate_grid_to_regions
from climate_toolbox_amd import aggregations as A

lat, lon, df = synth.realistic_segments()
T = 365
X = engine.synth_field(T, len(lat) * len(lon), seed=3, base=280.0, amp=60.0).cpu().numpy().reshape(T, len(lat), len(lon))
ds = minixr.Dataset({"tas": (("time", "lat", "lon"), X)}, coords={"lat": lat, "lon": lon})
for i in range(3):
    t0 = time.perf_counter()
    out = weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
    print("call %d: %.1f ms  -> %s" % (i, (time.perf_counter() - t0) * 1e3, out.tas.shape))
# breakdown of the host-side steps of a cached call (native label work)
import hashlib
t0 = time.perf_counter(); cell = A._resolve_cells(lat, lon, df["lat"].values, df["lon"].values); t1 = time.perf_counter()
u, c = A._factorize_labels(df["hierid"].values); t2 = time.perf_counter()
w = A._backup_fill(df["areawt"].values, df["areawt"].values); t3 = time.perf_counter()
h = hashlib.blake2b(digest_size=16)
for a in (cell, c, w): h.update(np.ascontiguousarray(a).tobytes())
t4 = time.perf_counter()
print("cell join %.1f ms, factorize (string labels) %.1f ms, backup fill %.1f ms, cache key %.1f ms"
      % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
