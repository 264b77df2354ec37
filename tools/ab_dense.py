#!/usr/bin/env python3
"""A/B timing of the dense MFMA kernel between two builds of libwagg on the SAME GPU (devices differ by
several percent, so numbers from different boxes cannot be compared): python tools/ab_dense.py libA.so libB.so
Each library runs the c2-dense contraction (T=365, G=1,036,800, R=24,378) in turn, twice."""
import ctypes as C
import json
import sys

import torch

T, G, R = 365, 720 * 1440, 24378
X = torch.empty((T, G), dtype=torch.float32, device="cuda").uniform_(250, 310)
out = torch.empty((T, R), dtype=torch.float32, device="cuda")
res = {}
for rep in range(2):
    for path in sys.argv[1:]:
        L = C.CDLL(path)
        vp = C.c_void_p
        L.wagg_dense_create_synth.argtypes = [C.c_int64, C.c_int32, C.c_uint32, C.POINTER(vp)]
        L.wagg_dense_apply_f32.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int, vp]
        L.wagg_dense_destroy.argtypes = [vp]
        L.wagg_profile_read.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
        h = vp()
        assert L.wagg_dense_create_synth(G, R, 2, C.byref(h)) == 0
        L.wagg_profile_enable(1)
        for _ in range(6):
            assert L.wagg_dense_apply_f32(h, X.data_ptr(), T, G, out.data_ptr(), R, 0, None) == 0
        torch.cuda.synchronize()
        buf = (C.c_float * 256)()
        n = C.c_int(0)
        L.wagg_profile_read(buf, 256, C.byref(n))
        ms = [buf[i] for i in range(n.value)][2:]
        res.setdefault(path, []).append(round(sum(ms) / len(ms), 3))
        L.wagg_profile_enable(0)
        L.wagg_dense_destroy(h)
        torch.cuda.empty_cache()
print(json.dumps(res))
