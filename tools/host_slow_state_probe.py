#!/usr/bin/env python3
"""One pass of the call order in which the half-rate state of the lines-only host path was seen in round 5 (DESIGN.md,
"host-resident fields"): plain lines-only calls, fused powers with whole rows page-locked in place, fused powers lines only,
plain lines-only calls again -- in a FRESH process whose HSA_ENABLE_SDMA is chosen on the command line (set before torch is
imported; a process that has touched the GPU is never re-launched).  Question (VERDICT r5 Next #5): does the state exist
when the runtime's copies run on the SDMA engines instead of blit kernels, and what does a lines-only call take there?

    python3 tools/host_slow_state_probe.py --sdma 0|1|default        # prints one JSON object
Under rocprofv3 set the variable on the command line in front of rocprofv3 (its preloaded library starts HSA first) and pass
--sdma keep."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--sdma", default="default", choices=["0", "1", "default", "keep"])
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
if a.sdma in ("0", "1"):
    os.environ["HSA_ENABLE_SDMA"] = a.sdma
elif a.sdma == "default":
    os.environ.pop("HSA_ENABLE_SDMA", None)

import numpy as np                                                                  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                        # noqa: E402,F401
from climate_toolbox_amd import _lib, engine, synth                                 # noqa: E402

lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R, T = len(lat) * len(lon), len(uniq), 365
plan = engine.SparsePlan(cell, code, w, G, R, row_len=len(lon))
X = (273.15 + 30 * np.random.default_rng(0).random((T, G), dtype=np.float32))
both = _lib.HOST_PIN | _lib.HOST_LINES
res = {"HSA_ENABLE_SDMA": os.environ.get("HSA_ENABLE_SDMA"), "legs": []}


def leg(name, fn, reps=a.reps, warm=2):
    for _ in range(warm):
        fn()
    _lib.host_stats(reset=True)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    st = _lib.host_stats()
    res["legs"].append({"leg": name, "ms": [round(t, 2) for t in ts], "median_ms": round(sorted(ts)[len(ts) // 2], 2),
                        "wait_copy_ms_per_call": round(st["lines_wait_copy_us"] / reps / 1e3, 2),
                        "wait_pack_ms_per_call": round(st["lines_wait_pack_us"] / reps / 1e3, 2),
                        "packed_gb_per_call": round(st["lines_h2d_bytes"] / reps / 1e9, 3),
                        "direct_h2d_gb_per_call": round(st["direct_h2d_bytes"] / reps / 1e9, 3),
                        "blocks_retired": st["blocks_retired"], "scratch_bytes": int(_lib.load().wagg_scratch_bytes())})


leg("plain_lines_first", lambda: plan.apply_host(X, flags=both))
leg("poly_whole_rows_pinned", lambda: plan.apply_poly_host(X, -273.15, 4, flags=_lib.HOST_PIN))
leg("poly_lines_only", lambda: plan.apply_poly_host(X, -273.15, 4, flags=both))
leg("plain_lines_last", lambda: plan.apply_host(X, flags=both))
leg("plain_whole_rows_pinned", lambda: plan.apply_host(X, flags=_lib.HOST_PIN))
print(json.dumps(res))
