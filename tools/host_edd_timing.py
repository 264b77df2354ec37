#!/usr/bin/env python3
"""The reference-named calls on HOST-resident c2-real fields that no row-block pipeline serves -- Snyder degree days (two
fields), the (lat, lon, time) layout of the reference's fixture -- end to end through the drop-in: what the upload costs now
(wagg_upload: page-locked in place, one DMA) against a pageable runtime copy of the same array.  Run on the GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_toolbox_amd import engine, synth

G = 720 * 1440
X = (280 + 20 * np.random.default_rng(0).standard_normal((365, G), dtype=np.float32))
res = {}


def med(fn, reps=5):
    fn(); fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(1e3 * sorted(ts)[len(ts) // 2], 2)


from climate_toolbox_amd import _lib
lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
R = len(uniq)
plan = engine.SparsePlan(cell, code, w, G, R, row_len=len(lon))
X -= np.float32(5.0)
Xmax = X + np.float32(9.0)


def upload_then_kernel():
    a, b = engine.upload(X), engine.upload(Xmax)
    return plan.apply_edd(a, b, [303.15]).cpu().numpy()


ORDER = (("edd_pipeline_lines_only_ms", _lib.HOST_PIN | _lib.HOST_LINES), ("edd_pipeline_whole_rows_ms", _lib.HOST_PIN),
         ("edd_pipeline_lines_only_again_ms", _lib.HOST_PIN | _lib.HOST_LINES))
if "upload_first" in sys.argv:
    res["edd_upload_then_kernel_ms"] = med(upload_then_kernel)
ref = None
for name, flags in ORDER:
    _lib.host_stats(reset=True)
    res[name] = med(lambda: plan.apply_edd_host(X, Xmax, [303.15], flags=flags))
    st = _lib.host_stats()
    res[name + "_stats"] = {"wait_pack_ms": round(st["lines_wait_pack_us"] / 7e3, 2), "wait_copy_ms": round(st["lines_wait_copy_us"] / 7e3, 2),
                            "packed_fraction": round(st["lines_h2d_bytes"] / 7 / (2 * X.nbytes), 3), "blocks_retired": st["blocks_retired"]}
    got = plan.apply_edd_host(X, Xmax, [303.15], flags=flags)
    ref = got if ref is None else ref
    assert np.array_equal(got, ref)
if "upload_first" not in sys.argv:
    res["edd_upload_then_kernel_ms"] = med(upload_then_kernel)
    assert np.array_equal(upload_then_kernel(), ref)
    _lib.host_stats(reset=True)
    res["edd_pipeline_lines_only_after_uploads_ms"] = med(lambda: plan.apply_edd_host(X, Xmax, [303.15], flags=_lib.HOST_PIN | _lib.HOST_LINES))
    st = _lib.host_stats()
    res["edd_pipeline_lines_only_after_uploads_ms_stats"] = {"wait_pack_ms": round(st["lines_wait_pack_us"] / 7e3, 2), "wait_copy_ms": round(st["lines_wait_copy_us"] / 7e3, 2), "blocks_retired": st["blocks_retired"]}
X64, Xmax64 = X.astype(np.float64), Xmax.astype(np.float64)
res["edd_f64_upload_then_kernel_ms"] = med(lambda: plan.apply_edd(engine.upload(X64), engine.upload(Xmax64), [303.15]).cpu().numpy(), reps=3)
res["edd_f64_pipeline_lines_only_ms"] = med(lambda: plan.apply_edd_host(X64, Xmax64, [303.15], flags=_lib.HOST_PIN | _lib.HOST_LINES), reps=3)
del X64, Xmax64
res["upload_1.5GB_ms"] = med(lambda: engine.upload(X))
res["pageable_runtime_copy_1.5GB_ms"] = med(lambda: torch.from_numpy(X).cuda())
print(json.dumps(res))
