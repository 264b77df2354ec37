#!/usr/bin/env python3
"""Randomised differential run of every C-ABI compute entry point against the oracle (GPU box).
usage: python tests/fuzz_gpu.py [n_cases] [seed]      -- prints one line per failing case, exit 1 on any"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from climate_toolbox_amd.engine import SparsePlan, DensePlan
from oracle import ref_numpy as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 0


N_LINES = 0


def rel_bad(got, ref, rtol, scale):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if got.shape != ref.shape:
        return "shape %r vs %r" % (got.shape, ref.shape)
    if not np.array_equal(np.isnan(got), np.isnan(ref)):
        return "NaN pattern differs at %d places" % (np.isnan(got) != np.isnan(ref)).sum()
    fin = np.isfinite(ref)
    inf = ~fin & ~np.isnan(ref)
    if not np.array_equal(got[inf], ref[inf]):
        return "inf values differ"
    err = np.abs(got[fin] - ref[fin])
    tol = rtol * np.maximum(np.abs(ref[fin]), scale)
    if (err > tol).any():
        return "max err/tol %.3g" % (err / tol).max()
    return None


def one_case(i, rng):
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    rtol = 1e-4 if dtype == np.float32 else 1e-6
    T = int(rng.choice([1, 2, 3, 7, 16, 63, 64, 65, 100, 129, 200, 366]))
    sc = int(os.environ.get("FUZZ_SCALE", "1"))                                        # larger grids on request
    nlat, nlon = int(rng.integers(1, 60 * sc)), int(rng.integers(1, 90 * sc))
    G = nlat * nlon
    R = int(rng.integers(1, 300))
    nseg = int(rng.integers(0, min(4 * G, 400000) + 2))
    cell = rng.integers(0, G, nseg).astype(np.int32)
    code = rng.integers(-1 if rng.random() < 0.3 else 0, R, nseg).astype(np.int32)     # some null labels
    if rng.random() < 0.5:
        # a COMPACT table (regions = blocks of the grid with holes, a few split cells): what the whole-line chunkings
        # are built for -- a scattered table above is declined by them (too many partial rows)
        bh, bw = int(rng.integers(1, 9)), int(rng.integers(1, 17))
        keep = np.flatnonzero(rng.random(G) < rng.uniform(0.3, 1.0))
        nbw = (nlon + bw - 1) // bw
        reg = ((keep // nlon) // bh) * nbw + (keep % nlon) // bw
        R = int(reg.max()) + 1 + int(rng.integers(0, 3)) if len(keep) else R           # (sometimes a region nobody maps to)
        extra = rng.choice(keep, min(len(keep), 40)) if len(keep) else keep
        cell = np.concatenate([keep, extra]).astype(np.int32)
        code = np.concatenate([reg, rng.integers(0, R, len(extra))]).astype(np.int32)
        if rng.random() < 0.3 and len(code):
            code[rng.integers(0, len(code), 3)] = -1
    if rng.random() < 0.4 and G > 700:                                                # a giant region
        n = int(rng.integers(300, min(G, 3000)))
        cell = np.concatenate([cell, rng.choice(G, n, replace=False).astype(np.int32)])
        code = np.concatenate([code, np.zeros(n, np.int32)])
    w = rng.uniform(0.05, 3.0, len(cell))
    w[rng.random(len(cell)) < 0.05] = np.nan
    w[rng.random(len(cell)) < 0.03] = 0.0
    X = (288.0 + 9.0 * rng.standard_normal((T, G))).astype(dtype)
    if rng.random() < 0.5:
        X[rng.integers(0, T, 5), rng.integers(0, G, 5)] = np.nan
    if rng.random() < 0.2:
        X[rng.integers(0, T), rng.integers(0, G)] = np.inf
    layout = "TG" if rng.random() < 0.7 else "GT"
    out_layout = "TR" if rng.random() < 0.7 else "RT"
    pad = int(rng.choice([0, 0, 1, 3, 4]))                                            # row stride > G / unaligned rows
    tag = "case %d: %s T=%d grid=%dx%d R=%d nseg=%d %s->%s pad=%d" % (i, dtype.__name__, T, nlat, nlon, R, len(cell), layout, out_layout, pad)
    plan = SparsePlan(cell, code, w, G, R, row_len=nlon)
    ref = O.agg_coded(X, cell, code, w, R)
    def dev(a):                                                                       # optionally padded rows
        a = a if layout == "TG" else np.ascontiguousarray(a.T)
        if pad:
            buf = torch.zeros((a.shape[0], a.shape[1] + pad), dtype=torch.from_numpy(a).dtype, device="cuda")
            buf[:, :a.shape[1]] = torch.from_numpy(a).cuda()
            return buf[:, :a.shape[1]]
        return torch.from_numpy(a).cuda()
    Xd = dev(X)
    fails = []
    got = plan.apply(Xd, layout=layout, out_layout=out_layout).cpu().numpy()
    got = got if out_layout == "TR" else got.T
    b = rel_bad(got, ref, rtol, 1.0)
    if b: fails.append("apply: " + b)
    from climate_toolbox_amd import _lib
    host_forms = layout == "TG" and out_layout == "TR"
    hflags = int(rng.choice([0, _lib.HOST_PIN, _lib.HOST_LINES, _lib.HOST_PIN | _lib.HOST_LINES, _lib.HOST_PIN | _lib.HOST_LINES | _lib.HOST_LINES_WHOLE]))
    if host_forms:                                            # the host forms of the same call (row-block pipeline; lines only
        _lib.host_stats(reset=True)                           #  when the field is large enough -- FUZZ_SCALE >= 6 -- and compact)
        gh = plan.apply_host(X, flags=hflags)
        b = rel_bad(gh, ref, rtol, 1.0)
        if b: fails.append("apply_host(flags %d): %s" % (hflags, b))
        if pad == 0 and not np.array_equal(gh, got, equal_nan=True): fails.append("apply_host(flags %d) differs from the device apply" % hflags)
        if _lib.host_stats()["lines_h2d_bytes"]: tag += " [lines]"
    kind = rng.integers(0, 3)
    if kind == 0:                                                                     # fused powers
        K = int(rng.integers(1, 6))
        gp = plan.apply_poly(Xd, -273.15, K, layout=layout, out_layout=out_layout).cpu().numpy()
        for p in range(1, K + 1):
            g = gp[p - 1] if out_layout == "TR" else gp[p - 1].T
            # sums of signed powers cancel: the error is relative to the size of the terms
            # (|y| ~ 15 +- 9 here), so |ref| is floored at 10^p
            b = rel_bad(g, O.agg_coded(O.tas_poly_values(X, p), cell, code, w, R), rtol, 10.0 ** p)
            if b: fails.append("poly p=%d: %s" % (p, b))
        if host_forms:
            gph = plan.apply_poly_host(X, -273.15, K, flags=hflags)
            if pad == 0 and not np.array_equal(gph, gp, equal_nan=True): fails.append("apply_poly_host(flags %d) differs from the device form" % hflags)
            for p in range(1, K + 1):
                b = rel_bad(gph[p - 1], O.agg_coded(O.tas_poly_values(X, p), cell, code, w, R), rtol, 10.0 ** p)
                if b: fails.append("poly_host p=%d: %s" % (p, b))
    elif kind == 1:                                                                   # degree days
        half = rng.uniform(0, 8, X.shape).astype(dtype)
        lo, hi = X - half, X + half
        thr = [float(rng.uniform(5, 35)) for _ in range(int(rng.integers(1, 7)))]        # 1..6: passes of four + a rest
        ge = plan.apply_edd(dev(lo), dev(hi), thr, offset=-273.15, layout=layout, out_layout=out_layout).cpu().numpy()
        ft = dtype
        for k, e in enumerate(thr):
            g = ge[k] if out_layout == "TR" else ge[k].T
            b = rel_bad(g, O.agg_coded(O.snyder_edd_values(lo + ft(-273.15), hi + ft(-273.15), e), cell, code, w, R), rtol, 0.05)
            if b: fails.append("edd e=%.2f: %s" % (e, b))
        if host_forms:
            geh = plan.apply_edd_host(lo, hi, thr, offset=-273.15, flags=hflags)
            if pad == 0 and not np.array_equal(geh, ge, equal_nan=True): fails.append("apply_edd_host(flags %d) differs from the device form" % hflags)
            for k, e in enumerate(thr):
                b = rel_bad(geh[k], O.agg_coded(O.snyder_edd_values(lo + ft(-273.15), hi + ft(-273.15), e), cell, code, w, R), rtol, 0.05)
                if b: fails.append("edd_host e=%.2f: %s" % (e, b))
    elif layout == "TG" and not np.isinf(X).any():            # dense family: the library's choice of form, or one pinned
        form = [None, "full", "tiles", "entries"][int(rng.integers(0, 4))]              # (round 5: the choice goes by estimated
        dp = DensePlan.from_segments(cell, code, w, G, R, dtype=dtype, form=form)       #  time, so small tables mostly take entry
        b = rel_bad(dp.apply(Xd.contiguous()).cpu().numpy(), ref, rtol, 1.0)            #  lists by themselves: pin the others)
        fname = ["full", "tiled", "entries"][dp.info["form"]]
        if b: fails.append("dense(%s, asked %s): %s" % (fname, form, b))
        # a tall ragged batch of the same rows (full row blocks + a shorter remainder: two launches), and a cloned plan
        T2 = int(rng.choice([500, 700, 1369, 1500]))
        idx = rng.integers(0, T, T2)
        Xt = Xd.contiguous()[torch.from_numpy(idx).cuda()]
        gt = dp.apply(Xt).cpu().numpy()
        b = rel_bad(gt, ref[idx], rtol, 1.0)
        if b: fails.append("dense(%s) tall batch T=%d: %s" % (fname, T2, b))
        if rng.random() < 0.4:
            rep = dp.replica(0)
            if not np.array_equal(rep.apply(Xt).cpu().numpy(), gt, equal_nan=True): fails.append("dense(%s): the clone's result differs" % fname)
            rep.close()
        dp.close()
        if rng.random() < 0.5 and len(cell):
            # the same table as CSR (rows = cells, columns ascending): the one-pass sort when nothing is dropped
            order = np.lexsort((code, cell))
            rowptr = np.zeros(G + 1, np.int64)
            np.add.at(rowptr, cell[order].astype(np.int64) + 1, 1)
            dc = DensePlan.from_csr(np.cumsum(rowptr), code[order], w[order], G, R, dtype=dtype, form=form)
            want_one_pass = int(not ((code < 0).any() or np.isnan(w).any()) and dc.info["one_pass_sort"] == 1)
            b = rel_bad(dc.apply(Xt).cpu().numpy(), ref[idx], rtol, 1.0)
            if b: fails.append("dense from CSR (one pass: %d): %s" % (dc.info["one_pass_sort"], b))
            if dc.info["one_pass_sort"] and ((code < 0).any() or np.isnan(w).any()): fails.append("one-pass sort taken with dropped rows")
            dc.close()
    plan.close()
    return tag, fails


def lines_case(i, rng):
    """FUZZ_LINES=1: fields large and tables compact enough for the lines-only host path (land masks with coasts, random
    grids and land fractions): the three host forms with WAGG_HOST_LINES against the device forms of the same kernels, bit
    for bit, and against the oracle."""
    from climate_toolbox_amd import _lib, synth
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    rtol = 1e-4 if dtype == np.float32 else 1e-6
    nlat, nlon = int(rng.integers(64, 200)), 4 * int(rng.integers(40, 160))
    G = nlat * nlon
    lat, lon, df = synth.realistic_segments(nlat=nlat, nlon=nlon, R=int(rng.integers(5, 900)), n_iso=5, seed=int(rng.integers(0, 1 << 30)),
                                            land_frac=float(rng.uniform(0.03, 0.35)), string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "popwt" if rng.random() < 0.5 else "areawt", "hierid")
    R = len(uniq)
    T = int(((64 << 20) // (G * np.dtype(dtype).itemsize) + 1) * rng.uniform(1.0, 1.6)) + int(rng.integers(0, 9))
    X = (288.0 + 9.0 * rng.standard_normal((T, G))).astype(dtype)
    X[rng.integers(0, T, 50), rng.integers(0, G, 50)] = np.nan
    if rng.random() < 0.3:
        X[rng.integers(0, T), cell[rng.integers(0, len(cell))]] = np.inf
    flags = _lib.HOST_LINES | (_lib.HOST_PIN if rng.random() < 0.7 else 0) | (_lib.HOST_LINES_WHOLE if rng.random() < 0.4 else 0)    # quads only / whole lines
    tag = "lines case %d: %s T=%d grid=%dx%d R=%d nseg=%d flags=%d" % (i, dtype.__name__, T, nlat, nlon, R, len(cell), flags)
    plan = SparsePlan(cell, code, w, G, R, row_len=nlon)
    fails = []
    Xd = torch.from_numpy(X).cuda()
    _lib.host_stats(reset=True)
    gh = plan.apply_host(X, flags=flags)
    took = _lib.host_stats()["lines_h2d_bytes"] > 0
    tag += " [lines]" if took else " [whole rows]"
    if not np.array_equal(gh, plan.apply(Xd).cpu().numpy(), equal_nan=True): fails.append("apply_host differs from the device apply")
    b = rel_bad(gh, O.agg_coded(X, cell, code, w, R), rtol, 1.0)
    if b: fails.append("apply_host: " + b)
    kind = int(rng.integers(0, 2))
    if kind == 0:
        K = int(rng.integers(1, 6))
        gp = plan.apply_poly_host(X, -273.15, K, flags=flags)
        if not np.array_equal(gp, plan.apply_poly(Xd, -273.15, K).cpu().numpy(), equal_nan=True): fails.append("apply_poly_host differs from the device form")
        b = rel_bad(gp[K - 1], O.agg_coded(O.tas_poly_values(X, K), cell, code, w, R), rtol, 10.0 ** K)
        if b: fails.append("poly_host p=%d: %s" % (K, b))
    else:
        half = rng.uniform(0, 8, X.shape).astype(dtype)
        lo, hi = X - half, X + half
        thr = [float(rng.uniform(5, 35)) for _ in range(int(rng.integers(1, 6)))]
        ge = plan.apply_edd_host(lo, hi, thr, offset=-273.15, flags=flags)
        gd = plan.apply_edd(torch.from_numpy(lo).cuda(), torch.from_numpy(hi).cuda(), thr, offset=-273.15).cpu().numpy()
        if not np.array_equal(ge, gd, equal_nan=True): fails.append("apply_edd_host differs from the device form")
        b = rel_bad(ge[0], O.agg_coded(O.snyder_edd_values(lo + dtype(-273.15), hi + dtype(-273.15), thr[0]), cell, code, w, R), rtol, 0.05)
        if b: fails.append("edd_host e=%.2f: %s" % (thr[0], b))
    plan.close()
    return tag, fails


def main():
    global one_case
    if os.environ.get("FUZZ_LINES"):
        one_case = lines_case
    rng = np.random.default_rng(SEED)
    bad = 0
    for i in range(N):
        try:
            tag, fails = one_case(i, rng)
        except Exception as e:                                                        # a crash is a failure too
            tag, fails = "case %d" % i, ["exception: %s" % traceback.format_exc().splitlines()[-1]]
        global N_LINES
        N_LINES += "[lines]" in tag
        if fails:
            bad += 1
            print(tag, "|", "; ".join(fails), flush=True)
        if i % 25 == 24:
            print("... %d cases, %d failing" % (i + 1, bad), flush=True)
    print("fuzz: %d cases, %d failing (%d of them took the lines-only host path)" % (N, bad, N_LINES))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
