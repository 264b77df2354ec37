#!/usr/bin/env python3
"""Writes the referenced line runs of the c2-real table (32-cell lines for fp32, 16-cell lines for fp64, optionally with
gaps of <= `gap` lines merged) to runs_<elem>_<gap>.bin for tools/micro/host_gather.cpp, and (round 6, VERDICT r5 Next #8) the
runs at finer granules -- runs_<elem>_g<granule bytes>.bin for 16 / 32 / 64 / 128-byte granules.  usage: host_gather_probe.py outdir"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climate_toolbox_amd import synth

out = sys.argv[1]
lat, lon, df = synth.realistic_segments(string_labels=False)
li = np.searchsorted(lat, df.lat.values); lj = np.searchsorted(lon, df.lon.values)
cell = np.unique(li * len(lon) + lj)
for elem, LINE in ((4, 32), (8, 16)):
    lines = np.unique(cell // LINE)
    for gap in (0, 1):
        brk = np.flatnonzero(np.diff(lines) > gap + 1)
        first = np.concatenate([[lines[0]], lines[brk + 1]])
        last = np.concatenate([lines[brk], [lines[-1]]])
        runs = np.stack([first * LINE, (last - first + 1) * LINE], axis=1).astype(np.int64)
        with open(os.path.join(out, "runs_%d_%d.bin" % (elem, gap)), "wb") as f:
            f.write(np.int64(len(runs)).tobytes()); f.write(runs.tobytes())

    for gb in (16, 32, 64, 128):
        LINE = gb // elem
        lines = np.unique(cell // LINE)
        brk = np.flatnonzero(np.diff(lines) > 1)
        first = np.concatenate([[lines[0]], lines[brk + 1]])
        last = np.concatenate([lines[brk], [lines[-1]]])
        runs = np.stack([first * LINE, (last - first + 1) * LINE], axis=1).astype(np.int64)
        with open(os.path.join(out, "runs_%d_g%d.bin" % (elem, gb)), "wb") as f:
            f.write(np.int64(len(runs)).tobytes()); f.write(runs.tobytes())
