import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from climate_toolbox_amd.engine import SparsePlan
from oracle import ref_numpy as O
rng = np.random.default_rng(5)
nlat, nlon, T = 64, 128, 130
G = nlat * nlon
cell = np.arange(G, dtype=np.int32); code = ((cell // nlon) // 8 * (nlon // 16) + (cell % nlon) // 16).astype(np.int32)
R = int(code.max()) + 1
w = rng.uniform(0.1, 1, G)
tmin = (rng.uniform(-20, 40, (T, G)) + 273.15); tmax = tmin + rng.uniform(0, 20, (T, G))
tmax[3, :50] = tmin[3, :50]            # w = 0
plan = SparsePlan(cell, code, w, G, R, row_len=nlon)
assert plan.info["lines"] == 7
thr = [5.0, 17.3, 31.0]
got = plan.apply_edd(torch.from_numpy(tmin).cuda(), torch.from_numpy(tmax).cuda(), thr, offset=-273.15).cpu().numpy()
worst = 0
for k, e in enumerate(thr):
    ref = O.agg_coded(O.snyder_edd_values(tmin - 273.15, tmax - 273.15, e), cell, code, w, R)
    err = np.abs(got[k] - ref) / np.maximum(np.abs(ref), 1e-3)
    worst = max(worst, err.max())
print("fp64 degree days vs oracle: max rel err %.3g" % worst)
