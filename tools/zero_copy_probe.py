#!/usr/bin/env python3
"""Host-resident c2-real field, two ways: the row-block pipeline of wagg_apply_host_ex_* (every byte of X
crosses PCIe) against ONE apply whose X pointer is the device alias of the page-locked host array (the gather kernel
then pulls only the 128-byte lines the table references: 64 % of the field).  Go / no-go measurement for a
"lines only" host path.  Run on the GPU box; prints one JSON object."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_toolbox_amd import _lib, engine, synth

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]

lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R = len(lat) * len(lon), len(uniq)
res = {}
L = _lib.load()
for dtype, fn in ((np.float32, L.wagg_apply_f32), (np.float64, L.wagg_apply_f64)):
    T = 365
    X = (280 + 20 * np.random.default_rng(0).standard_normal((T, G), dtype=np.float32)).astype(dtype)
    name = np.dtype(dtype).name
    for pname, pflags in (("lines", 0), ("quads", _lib.PLAN_NO_LINES)):
        plan = engine.SparsePlan(cell, code, w, G, R, row_len=len(lon), flags=pflags)
        ref = plan.apply_host(X, flags=_lib.HOST_PIN)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); plan.apply_host(X, flags=_lib.HOST_PIN); ts.append(time.perf_counter() - t0)
        res["%s_%s_pipeline_ms" % (name, pname)] = round(1e3 * sorted(ts)[len(ts) // 2], 2)
        # zero copy: register (mapped), alias, one apply, result back through a pinned torch buffer
        out_d = torch.empty((T, R), dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
        out_h = torch.empty((T, R), dtype=out_d.dtype, pin_memory=True)
        reg, app, tot = [], [], []
        for _ in range(5):
            t0 = time.perf_counter()
            rc = hip.hipHostRegister(X.ctypes.data, X.nbytes, 1 | 2)          # portable | mapped
            assert rc == 0, rc
            dp = C.c_void_p()
            assert hip.hipHostGetDevicePointer(C.byref(dp), X.ctypes.data, 0) == 0
            t1 = time.perf_counter()
            _lib.check(fn(plan._h, dp, T, G, _lib.LAYOUT_TG, C.c_void_p(out_d.data_ptr()), R, _lib.OUT_TR,
                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "apply")
            out_h.copy_(out_d, non_blocking=True)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            assert hip.hipHostUnregister(X.ctypes.data) == 0
            t3 = time.perf_counter()
            reg.append((t1 - t0) + (t3 - t2)); app.append(t2 - t1); tot.append(t3 - t0)
        med = lambda v: round(1e3 * sorted(v)[len(v) // 2], 2)
        res["%s_%s_zero_copy_ms" % (name, pname)] = {"register+unregister": med(reg), "apply+result": med(app), "total": med(tot)}
        res["%s_%s_equal" % (name, pname)] = bool(np.array_equal(out_h.numpy(), ref))
        plan.close()
    del X
print(json.dumps(res))
