"""Where a repeated call of the reference-named functions on a device-resident c2-real field spends its host time (bare
DataFrame: the table is re-fingerprinted by content on every call).  GPU box."""
import os, sys, time, collections
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from climate_toolbox_amd import engine, minixr, synth, aggregations as A
lat, lon, df = synth.realistic_segments()
T = 365
Xd = engine.synth_field(T, len(lat) * len(lon), seed=3, base=280.0, amp=60.0).reshape(T, len(lat), len(lon))
ds = minixr.Dataset({"tas": (("time", "lat", "lon"), Xd)}, coords={"lat": lat, "lon": lon})
acc = collections.defaultdict(float)
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t0
    setattr(obj, name, g)
for n in ("_backup_fill", "_factorize_labels", "_plan_for", "_to_host", "_flatten_for_device", "_resolve_cells", "_fingerprint"):
    wrap(A, n)
wrap(engine.SparsePlan, "apply"); wrap(engine.SparsePlan, "status"); wrap(A.ReindexedDataset, "_cell_index")
out = None
for i in range(int(os.environ.get("N_CALLS", "14"))):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    re = A._reindex_spatial_data_to_regions(ds, df); t1 = time.perf_counter()
    o2 = A._aggregate_reindexed_data_to_regions(re, "tas", "areawt", "hierid", df); t2 = time.perf_counter()
    out = o2; del o2, re
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if i >= 8 and i % int(os.environ.get("EVERY", "1")) == 0: print("reindex %.2f aggregate %.2f total %.2f ms | " % ((t1-t0)*1e3, (t2-t1)*1e3, (t4-t0)*1e3) + " ".join("%s=%.2f" % (k, v * 1e3) for k, v in sorted(acc.items())))
