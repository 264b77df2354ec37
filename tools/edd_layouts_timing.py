import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from climate_toolbox_amd import synth
from climate_toolbox_amd.engine import SparsePlan, synth_field
lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "popwt", "hierid")
G, R, T = len(lat) * len(lon), len(uniq), 365
plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
res = {}
for dt in ("float64", "float32"):
    tmin = synth_field(T, G, 7, 288.0, 30.0, dtype=dt)
    tmax = tmin + synth_field(T, G, 8, 6.0, 10.0, dtype=dt)
    for lay in ("TG", "GT"):
        a, b = (tmin, tmax) if lay == "TG" else (tmin.t().contiguous(), tmax.t().contiguous())
        for K in (1, 3):
            thr = [10.0, 20.0, 30.0][:K]
            out = torch.empty((K, T, R), dtype=tmin.dtype, device="cuda")
            for _ in range(2): plan.apply_edd(a, b, thr, offset=-273.15, layout=lay, out=out)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): plan.apply_edd(a, b, thr, offset=-273.15, layout=lay, out=out)
            torch.cuda.synchronize(); res["%s_%s_K%d_ms" % (dt, lay, K)] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
print(json.dumps(res))
