#!/usr/bin/env python3
"""Drives the host half of libwagg (label join, backup fill, factorisation, the sparse plan builder
and the dense-from-segments builder) and the C oracle under AddressSanitizer + UBSan (SURVEY.md
section 5).  Run through tests/test_sanitize.py, which builds `make -C climate_toolbox_amd/csrc
hostsan` / `make -C oracle san` and preloads the sanitizer runtime; ctypes only (no torch).

Exit status 0 = every call returned what it should and the sanitizers stayed silent (they abort
the process otherwise)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "climate_toolbox_amd", "lib", "libwagg_hostsan.so"))
O = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libwagg_oracle_san.so"))
L.wagg_last_error.restype = C.c_char_p
f64p, i32p, i64p, u8p = (C.POINTER(t) for t in (C.c_double, C.c_int32, C.c_int64, C.c_uint8))


def p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


rng = np.random.default_rng(0)
# --- label join (exact match, KeyError row) ---------------------------------------------------
lat, lon = np.arange(-89.875, 90, 2.0), np.arange(0.125, 360.0, 2.0)
n = 5000
sa, so = rng.choice(lat, n), rng.choice(lon, n)
cell = np.empty(n, np.int32)
bad = C.c_int64(-1)
assert L.wagg_resolve_cells(p(lat, C.c_double), C.c_int64(len(lat)), p(lon, C.c_double), C.c_int64(len(lon)),
                            p(sa, C.c_double), p(so, C.c_double), C.c_int64(n), 0, p(cell, C.c_int32), C.byref(bad)) == 0
assert (lat[cell // len(lon)] == sa).all() and (lon[cell % len(lon)] == so).all()
so2 = so.copy(); so2[77] = 0.126
assert L.wagg_resolve_cells(p(lat, C.c_double), C.c_int64(len(lat)), p(lon, C.c_double), C.c_int64(len(lon)),
                            p(sa, C.c_double), p(so2, C.c_double), C.c_int64(n), 1, p(cell, C.c_int32), C.byref(bad)) == -6
assert bad.value == 77
# --- backup fill / relabel -----------------------------------------------------------------------
w = rng.standard_normal(n); w[::7] = np.nan
bk = rng.uniform(0.1, 1, n)
we = np.empty(n)
assert L.wagg_backup_fill(p(w, C.c_double), p(bk, C.c_double), C.c_int64(n), p(we, C.c_double)) == 0
np.testing.assert_array_equal(we, np.where(w > 0, w, bk))
x = np.array([180.125, 1.0, 180.125, -5.0])
assert L.wagg_relabel(p(x, C.c_double), C.c_int64(4), C.c_double(180.125), C.c_double(-179.875)) == 0
assert list(x) == [-179.875, 1.0, -179.875, -5.0]
# --- factorisation -----------------------------------------------------------------------------
lab = rng.integers(-50, 50, n).astype(np.int64)
null = (rng.random(n) < 0.05).astype(np.uint8)
codes, uniq, nu = np.empty(n, np.int32), np.empty(n, np.int64), C.c_int64(0)
assert L.wagg_factorize_i64(p(lab, C.c_int64), p(null, C.c_uint8), C.c_int64(n), p(codes, C.c_int32), p(uniq, C.c_int64), C.byref(nu)) == 0
u = np.unique(lab[null == 0])
np.testing.assert_array_equal(uniq[:nu.value], u)
assert (codes[null == 1] == -1).all() and (u[codes[null == 0]] == lab[null == 0]).all()
strs = np.array(["R%04d" % k for k in rng.integers(0, 300, n)], dtype="S5")
rows = np.empty(n, np.int64)
assert L.wagg_factorize_bytes(strs.ctypes.data_as(C.c_char_p), C.c_int64(5), None, C.c_int64(n), p(codes, C.c_int32), p(rows, C.c_int64), C.byref(nu)) == 0
np.testing.assert_array_equal(strs[rows[:nu.value]], np.unique(strs))
# --- plan builders: all the host work (coalesce, band ordering, chunk packing, descriptors) runs
# before the first device call; without a GPU the upload then fails with a status, not a crash ---
G, R = 720 * 90, 1500
for nseg, giant in ((0, 0), (1, 0), (40000, 0), (30000, 3)):
    ci = rng.integers(0, G, nseg).astype(np.int32)
    rc = rng.integers(-1, R, nseg).astype(np.int32)
    if giant:
        rc[: nseg // 2] = rng.integers(0, giant, nseg // 2)          # a few regions with > 64 quads
    ww = rng.uniform(-0.1, 1, nseg); ww[::11] = np.nan
    h = C.c_void_p()
    rcode = L.wagg_plan_create(p(ci, C.c_int32), p(rc, C.c_int32), p(ww, C.c_double), C.c_int64(nseg), C.c_int64(G),
                               C.c_int32(R), C.c_int64(720), 0, C.byref(h))
    assert rcode in (0, -2, -4), (rcode, L.wagg_last_error())
    if rcode == 0:
        L.wagg_plan_destroy(h)
    h2 = C.c_void_p()
    rcode = L.wagg_dense_create_from_segments(p(ci, C.c_int32), p(rc, C.c_int32), p(ww, C.c_double), C.c_int64(nseg),
                                              C.c_int64(G), C.c_int32(R), 0, C.byref(h2))
    assert rcode in (0, -1, -2, -3, -4), (rcode, L.wagg_last_error())
    if rcode == 0:
        L.wagg_dense_destroy(h2)
    # the CSR form of the same table: the host validates the row offsets, then the upload fails with a status
    order = np.argsort(ci, kind="stable")
    rowptr = np.zeros(G + 1, np.int64); np.add.at(rowptr, ci.astype(np.int64) + 1, 1); rowptr = np.cumsum(rowptr)
    col, val = np.ascontiguousarray(rc[order]), np.ascontiguousarray(ww[order])
    h3 = C.c_void_p()
    rcode = L.wagg_dense_create_from_csr(p(rowptr, C.c_int64), p(col, C.c_int32), p(val, C.c_double), C.c_int64(G), C.c_int32(R), 0, C.byref(h3))
    assert rcode in (0, -1, -2, -3, -4), (rcode, L.wagg_last_error())
    if rcode == 0:
        L.wagg_dense_destroy(h3)
    if nseg > 1:
        bad = rowptr.copy(); bad[G // 2] = bad[G // 2 + 1] + 1
        assert L.wagg_dense_create_from_csr(p(bad, C.c_int64), p(col, C.c_int32), p(val, C.c_double), C.c_int64(G), C.c_int32(R), 0, C.byref(h3)) == -1
# compact regions on a grid whose rows are not a whole number of 32-cell lines: the whole-line plan (chunks of eight
# lines per column strip, partial rows) and, with WAGG_PLAN_NO_LINES = 4, the region-shaped chunks
nlat, nlon = 61, 100
cells = np.flatnonzero(rng.random(nlat * nlon) < 0.6).astype(np.int32)
reg = ((cells // nlon) // 5 * 20 + (cells % nlon) // 5).astype(np.int32)
dup = rng.integers(0, len(cells), 700)
ci = np.concatenate([cells, cells[dup]]); rc = np.concatenate([reg, (reg[dup] + 1) % (reg.max() + 1)])
ww = rng.uniform(0.1, 1, len(ci))
for fl in (0, 4, 16, 20):                  # 16 = WAGG_PLAN_SERIAL_BUILD: the chunkings one after the other on this thread (what the
    h = C.c_void_p()                       # library falls back to when it cannot start its worker threads)
    rcode = L.wagg_plan_create(p(ci, C.c_int32), p(rc, C.c_int32), p(ww, C.c_double), C.c_int64(len(ci)), C.c_int64(nlat * nlon),
                               C.c_int32(int(reg.max()) + 1), C.c_int64(nlon), fl, C.byref(h))
    assert rcode in (0, -2, -4), (rcode, L.wagg_last_error())
    if rcode == 0:
        L.wagg_plan_destroy(h)
# out-of-range rows are rejected, not read past
ci = np.array([G], np.int32); rc = np.array([0], np.int32); ww = np.array([1.0])
h = C.c_void_p()
assert L.wagg_plan_create(p(ci, C.c_int32), p(rc, C.c_int32), p(ww, C.c_double), C.c_int64(1), C.c_int64(G), C.c_int32(R),
                          C.c_int64(0), 0, C.byref(h)) == -1
# --- the C oracle --------------------------------------------------------------------------------
T, G, R, nseg = 7, 500, 20, 3000
X = rng.standard_normal((T, G)); X[2, 5] = np.nan
ci = rng.integers(0, G, nseg).astype(np.int32); rc = rng.integers(-1, R, nseg).astype(np.int32)
ww = rng.uniform(0, 1, nseg); ww[::13] = np.nan
out = np.empty((T, R))
O.wagg_oracle_segments_f64.restype = C.c_int
assert O.wagg_oracle_segments_f64(p(X, C.c_double), C.c_int64(T), C.c_int64(G), 0, p(ci, C.c_int32), p(rc, C.c_int32),
                                  p(ww, C.c_double), C.c_int64(nseg), C.c_int64(G), C.c_int32(R), p(out, C.c_double)) == 0
Xf = X.astype(np.float32)
outd = np.empty((T, 16))
assert O.wagg_oracle_dense_synth_f32(p(Xf, C.c_float), C.c_int64(T), C.c_int64(G), C.c_int64(0), C.c_int64(G), C.c_int64(64),
                                     C.c_int64(8), C.c_int64(16), C.c_uint32(2), p(outd, C.c_double)) == 0
print("hostsan ok")
sys.exit(0)
