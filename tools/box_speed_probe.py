#!/usr/bin/env python3
"""How fast is THIS box on the HBM-latency-bound segment-table kernel?  (MI355X boxes of the pool differ by ~10 % on it and
by < 1 % on the MFMA kernels: profiles/README.md, round 6.)  Prints the median dispatch time of 300 c2-real fp32 applies
(hipExtLaunchKernel events) and exits 0 when it is below the threshold given (ms), 1 otherwise."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climate_toolbox_amd import engine, synth

thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.225
lat, lon, df = synth.realistic_segments(string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G, R, T = len(lat) * len(lon), len(uniq), 365
plan = engine.SparsePlan(cell, code, w, G, R, row_len=len(lon))
X = engine.synth_field(T, G, seed=1000, base=280.0, amp=60.0, dtype="float32")
out = torch.empty((T, R), dtype=torch.float32, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    plan.apply(X, out=out)
    torch.cuda.synchronize()
engine.profile_enable(True)
for _ in range(300):
    plan.apply(X, out=out)
torch.cuda.synchronize()
ms = sorted(engine.profile_read())
engine.profile_enable(False)
med = ms[len(ms) // 2]
import subprocess
try:
    smi = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showtemp", "--showmemuse"], capture_output=True, text=True, timeout=30).stdout
    print("\n".join(l for l in smi.splitlines() if any(k in l for k in ("mclk", "sclk", "fclk", "Temperature", "memory use", "Memory Activity"))))
except Exception as e:
    print("rocm-smi:", e)
print("c2-real kernel median %.4f ms (min %.4f) -- threshold %.4f" % (med, ms[0], thr))
sys.exit(0 if med < thr else 1)
