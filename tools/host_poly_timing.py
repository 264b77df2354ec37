#!/usr/bin/env python3
"""tas_poly-then-aggregate on a HOST-resident c2-real field (powers 1..4): the row-block pipeline with the powers fused
(wagg_apply_poly_host_*, whole rows / lines only) against what the drop-in did before -- one pageable copy of the whole field,
then the fused kernel on the device.  Prints one JSON object.  Run on the GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_toolbox_amd import _lib, engine, synth

lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R, T = len(lat) * len(lon), len(uniq), 365
plan = engine.SparsePlan(cell, code, w, G, R, row_len=len(lon))
X = (273.15 + 30 * np.random.default_rng(0).random((T, G), dtype=np.float32))
res = {}


def med(fn, reps=5):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t0)
    return round(1e3 * sorted(ts)[len(ts) // 2], 2), out


def copy_then_kernel():
    Xd = torch.from_numpy(X).cuda()                      # pageable -> device, as the drop-in's _to_device did
    out = plan.apply_poly(Xd, -273.15, 4).cpu().numpy()
    return out


VARIANTS = "variants" in sys.argv          # (with these calls in between the slow state of docs/HISTORY.md (f) did not show)
OUT_KEEP = np.empty((T, R), dtype=np.float32)


def c_call(flags):
    import ctypes as C
    rc = _lib.load().wagg_apply_host_ex_f32(plan._h, X.ctypes.data_as(C.c_void_p), T, G, 0, OUT_KEEP.ctypes.data_as(C.c_void_p), R, 0, flags)
    assert rc == 0


def plain(tag):
    if VARIANTS and (tag.startswith("after_") or tag == "last"):
        res["variants_" + tag] = {"staged_result_ms": med(lambda: plan.apply_host(X, flags=_lib.HOST_LINES))[0],
                                  "reused_result_array_ms": med(lambda: c_call(_lib.HOST_PIN | _lib.HOST_LINES))[0],
                                  "whole_rows_pinned_ms": med(lambda: plan.apply_host(X, flags=_lib.HOST_PIN))[0]}
    _lib.host_stats(reset=True)
    ms, _ = med(lambda: plan.apply_host(X, flags=_lib.HOST_PIN | _lib.HOST_LINES))
    st = _lib.host_stats()
    res["plain_aggregation_lines_only_" + tag] = {"ms": ms, "wait_pack_ms": round(st["lines_wait_pack_us"] / 7e3, 2),
                                                  "wait_copy_ms": round(st["lines_wait_copy_us"] / 7e3, 2),
                                                  "packed_fraction_of_x": round(st["lines_h2d_bytes"] / 7 / X.nbytes, 3),
                                                  "blocks_retired": st["blocks_retired"]}


plain("first")
res["copy_then_kernel_ms"], ref = med(copy_then_kernel)
plain("after_torch_pageable_copies")
for name, flags in (("pipeline_whole_rows_ms", _lib.HOST_PIN), ("pipeline_lines_only_ms", _lib.HOST_PIN | _lib.HOST_LINES)):
    res[name], out = med(lambda: plan.apply_poly_host(X, -273.15, 4, flags=flags))
    assert np.array_equal(out, ref)
    del out
    res["scratch_bytes_after_" + name] = _lib.load().wagg_scratch_bytes()
    plain("after_" + name)
plain("last")
_lib.load().wagg_release_scratch()
plain("last_after_release_scratch")
print(json.dumps(res))
