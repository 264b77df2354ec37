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
    for name, flags in (("whole", _lib.HOST_WHOLE), ("blocks_pageable", 0), ("blocks_pinned", _lib.HOST_PIN),
                        ("lines_only", _lib.HOST_LINES | _lib.HOST_PIN)):
        _lib.host_stats(reset=True)
        dt, out = timed(lambda: plan.apply_host(X, flags=flags), reps=5)
        ref = out if ref is None else ref
        assert np.array_equal(out, ref)
        res["sparse_T%d_%s" % (T, name)] = {"ms": round(dt * 1e3, 2), "x_gb_per_s": round(X.nbytes / dt / 1e9, 1)}
        if name == "lines_only":
            st = _lib.host_stats()
            res["sparse_T%d_%s" % (T, name)].update(packed_fraction=round(st["lines_h2d_bytes"] / 6 / X.nbytes, 3),
                                                     wait_pack_ms=round(st["lines_wait_pack_us"] / 6e3, 2),
                                                     wait_copy_ms=round(st["lines_wait_copy_us"] / 6e3, 2))
            ts = []
            out_keep = np.empty_like(out)
            for _ in range(7):                                 # the C call alone, result buffer reused
                t0 = time.perf_counter()
                rc = _lib.load().wagg_apply_host_ex_f32(plan._h, X.ctypes.data_as(__import__("ctypes").c_void_p), T, G, 0,
                                                        out_keep.ctypes.data_as(__import__("ctypes").c_void_p), R, 0, flags)
                ts.append(time.perf_counter() - t0)
                assert rc == 0
            res["sparse_T%d_%s" % (T, name)]["c_call_ms_sorted"] = [round(1e3 * t, 2) for t in sorted(ts)]
    if T == 365:                                               # the fp64 field of c3 (3 GB)
        X64 = X.astype(np.float64)
        ref = None
        for name, flags in (("blocks_pinned", _lib.HOST_PIN), ("lines_only", _lib.HOST_LINES | _lib.HOST_PIN)):
            dt, out = timed(lambda: plan.apply_host(X64, flags=flags), reps=5)
            ref = out if ref is None else ref
            assert np.array_equal(out, ref)
            res["sparse_f64_T%d_%s" % (T, name)] = {"ms": round(dt * 1e3, 2), "x_gb_per_s": round(X64.nbytes / dt / 1e9, 1)}
        del X64
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
