"""Times the drop-in on HOST-resident data (the PCIe-inclusive number quoted in DESIGN.md):
c2-real, T=365, 0.25-degree grid, fp32.  usage: python tools/host_path_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from climate_toolbox_amd import engine, synth

lat, lon, df = synth.realistic_segments()
cell, codes, w_eff, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, T = len(lat) * len(lon), 365
plan = engine.SparsePlan(cell, codes, w_eff, G, len(uniq), row_len=len(lon))
X = engine.synth_field(T, G, seed=3, base=280.0, amp=60.0).cpu().numpy()
for name, fn in (("pageable .cuda() + apply + .cpu()", lambda: plan.apply(torch.from_numpy(X).cuda()).cpu().numpy()),
                 ("wagg_apply_host_f32 (C-ABI, hipMemcpy)", lambda: plan.apply_host(X))):
    fn()
    t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
    print("%-42s %.1f ms  (%.1f GB/s of X)" % (name, dt * 1e3, X.nbytes / dt * 1e-9))
