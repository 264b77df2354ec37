#!/usr/bin/env python3
"""Host-resident data through the C-ABI: whole-field copy vs the row-block pipeline of
wagg_apply_host_ex_* / wagg_dense_apply_host_*, pageable vs page-locked in place.  Prints the wall
time per call and the PCIe-inclusive rate of X (GB/s).  Run on the GPU box."""
import json, sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from climate_toolbox_amd import _lib, engine, synth

lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R = len(lat) * len(lon), len(uniq)
plan = engine.SparsePlan(cell, code, w, G, R, row_len=len(lon))
res = {}


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


for T in (365, 1369):
    X = (280 + 20 * np.random.default_rng(0).standard_normal((T, G), dtype=np.float32))
    ref = None
    for name, flags in (("whole", _lib.HOST_WHOLE), ("blocks_pageable", 0), ("blocks_pinned", _lib.HOST_PIN)):
        dt, out = timed(lambda: plan.apply_host(X, flags=flags))
        ref = out if ref is None else ref
        assert np.array_equal(out, ref)
        res["sparse_T%d_%s" % (T, name)] = {"ms": round(dt * 1e3, 2), "x_gb_per_s": round(X.nbytes / dt / 1e9, 1)}
    del X
if len(sys.argv) > 1 and sys.argv[1] == "dense":
    dplan = engine.DensePlan.synth(G, R, seed=2)               # 101 GB of W in HBM
    T = 1369
    X = (280 + 20 * np.random.default_rng(0).standard_normal((T, G), dtype=np.float32))
    ref = None
    for name, flags in (("whole", _lib.HOST_WHOLE), ("blocks_pageable", 0), ("blocks_pinned", _lib.HOST_PIN)):
        dt, out = timed(lambda: dplan.apply_host(X, flags=flags), reps=2)
        ref = out if ref is None else ref
        np.testing.assert_allclose(out, ref, rtol=2e-5)
        res["dense_T%d_%s" % (T, name)] = {"ms": round(dt * 1e3, 1), "x_gb_per_s": round(X.nbytes / dt / 1e9, 1)}
print(json.dumps(res))
