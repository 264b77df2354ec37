"""ctypes loader for oracle/_build/libwagg_oracle.so.  TEST INFRASTRUCTURE ONLY (see ref_numpy.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libwagg_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "wagg_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        i32p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        for name, xp in (("wagg_oracle_segments_f32", C.POINTER(C.c_float)), ("wagg_oracle_segments_f64", f64p),
                         ("wagg_oracle_segments_omp_f32", C.POINTER(C.c_float)), ("wagg_oracle_segments_omp_f64", f64p)):
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = [xp, C.c_int64, C.c_int64, C.c_int, i32p, i32p, f64p, C.c_int64,
                           C.c_int64, C.c_int32, f64p]
        L.wagg_oracle_dense_synth_f32.restype = C.c_int
        L.wagg_oracle_dense_synth_f32.argtypes = [C.POINTER(C.c_float)] + [C.c_int64] * 7 + [
            C.c_uint32, f64p]
        L.wagg_oracle_dense_synth2_f32.restype = C.c_int
        L.wagg_oracle_dense_synth2_f32.argtypes = [C.POINTER(C.c_float)] + [C.c_int64] * 7 + [
            C.c_uint32, C.c_double, C.c_int, f64p]
        L.wagg_oracle_dense_synth_cols_f32.restype = C.c_int
        L.wagg_oracle_dense_synth_cols_f32.argtypes = [C.POINTER(C.c_float)] + [C.c_int64] * 4 + [
            C.POINTER(C.c_int64), C.c_int64, C.c_uint32, C.c_double, C.c_int, f64p]
        L.wagg_oracle_hash_u01.restype = C.c_float
        L.wagg_oracle_hash_u01.argtypes = [C.c_uint64, C.c_uint32]
        L.wagg_oracle_threads.restype = C.c_int
        L.wagg_oracle_synth_csr.restype = C.c_int64
        L.wagg_oracle_synth_csr.argtypes = [C.c_int64, C.c_int64, C.c_uint32, C.c_double, C.c_int, C.c_int64, C.c_int64,
                                            C.POINTER(C.c_int64), i32p, f64p, C.c_int64]
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def segments(X, cell_idx, region_code, w_eff, R, layout="TG", threaded=False, out=None):
    """fp64 (T, R) result of the faithful single-threaded C restatement (``threaded``: the same sums with the
    timesteps dealt to OpenMP threads -- the best-effort CPU leg of bench.py; identical bits).  ``out``: a (T, R) float64
    array to write into (repeated timing calls reuse one touched buffer instead of faulting 70 MB of fresh pages in)."""
    L = lib()
    X = np.ascontiguousarray(X)
    assert X.ndim == 2 and X.dtype in (np.float32, np.float64)
    if layout == "TG":
        T, G = X.shape
        lay = 0
    else:
        G, T = X.shape
        lay = 1
    ldx = X.shape[1]
    ci = np.ascontiguousarray(cell_idx, dtype=np.int32)
    rc = np.ascontiguousarray(region_code, dtype=np.int32)
    we = np.ascontiguousarray(w_eff, dtype=np.float64)
    if out is None:
        out = np.empty((T, R), dtype=np.float64)
    assert out.shape == (T, R) and out.dtype == np.float64 and out.flags.c_contiguous
    fn = getattr(L, "wagg_oracle_segments_%s%s" % ("omp_" if threaded else "", "f32" if X.dtype == np.float32 else "f64"))
    ct = C.c_float if X.dtype == np.float32 else C.c_double
    rcode = fn(_p(X, ct), T, ldx, lay, _p(ci, C.c_int32), _p(rc, C.c_int32), _p(we, C.c_double),
               len(ci), G, R, _p(out, C.c_double))
    if rcode != 0:
        raise RuntimeError("wagg_oracle_segments failed: %d" % rcode)
    return out


def dense_synth(X, g0, Gw, R_total, r0, Rw, seed):
    L = lib()
    X = np.ascontiguousarray(X, dtype=np.float32)
    T, ldx = X.shape
    out = np.empty((T, Rw), dtype=np.float64)
    rcode = L.wagg_oracle_dense_synth_f32(_p(X, C.c_float), T, ldx, g0, Gw, R_total, r0, Rw,
                                          seed, _p(out, C.c_double))
    if rcode != 0:
        raise RuntimeError("wagg_oracle_dense_synth_f32 failed: %d" % rcode)
    return out


def dense_synth_sparse(X, g0, Gw, R_total, r0, Rw, seed, fill=1.0, blocklocal=False):
    """Column window [r0, r0 + Rw) of the synthetic weights with a fraction ``fill`` of non-zeros
    (c5: uniform-random, or block-local with ``blocklocal``), kept pairs only, fp64 accumulation."""
    L = lib()
    X = np.ascontiguousarray(X, dtype=np.float32)
    T, ldx = X.shape
    out = np.empty((T, Rw), dtype=np.float64)
    rcode = L.wagg_oracle_dense_synth2_f32(_p(X, C.c_float), T, ldx, g0, Gw, R_total, r0, Rw, seed,
                                           float(fill), 1 if blocklocal else 0, _p(out, C.c_double))
    if rcode != 0:
        raise RuntimeError("wagg_oracle_dense_synth2_f32 failed: %d" % rcode)
    return out


def dense_synth_cols(X, G, R_total, cols, seed, fill=1.0, blocklocal=False):
    """The synthetic dense / c5 weights contracted for an arbitrary list of regions ``cols`` over all G
    cells: (T, len(cols)) fp64.  One short window per column tile of a full-size result costs seconds."""
    L = lib()
    X = np.ascontiguousarray(X, dtype=np.float32)
    T, ldx = X.shape
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    out = np.empty((T, len(cols)), dtype=np.float64)
    rcode = L.wagg_oracle_dense_synth_cols_f32(_p(X, C.c_float), T, ldx, G, R_total, _p(cols, C.c_int64), len(cols),
                                               seed, float(fill), 1 if blocklocal else 0, _p(out, C.c_double))
    if rcode != 0:
        raise RuntimeError("wagg_oracle_dense_synth_cols_f32 failed: %d" % rcode)
    return out


def synth_csr(G, R, seed, fill, blocklocal=False, g0=0, g1=None):
    """The c5 weight table (rows [g0, g1) of it) as host CSR arrays: (rowptr int64, col int32, val float64)."""
    L = lib()
    g1 = G if g1 is None else g1
    rowptr = np.empty(g1 - g0 + 1, dtype=np.int64)
    # arrays a little above the expected size, one call (counting pass + filling pass); exact-size retry if that was short
    width = 256 if blocklocal else R
    cap = int(1.02 * fill * width * (g1 - g0)) + 4096
    for attempt in range(2):
        col = np.empty(cap, dtype=np.int32)
        val = np.empty(cap, dtype=np.float64)
        got = L.wagg_oracle_synth_csr(G, R, seed, float(fill), 1 if blocklocal else 0, g0, g1, _p(rowptr, C.c_int64),
                                      _p(col, C.c_int32), _p(val, C.c_double), cap)
        if got >= 0:
            return rowptr, col[:got], val[:got]
        if got != -2 or attempt:
            raise RuntimeError("wagg_oracle_synth_csr failed: %d" % got)
        cap = int(rowptr[-1])


def threads():
    return int(lib().wagg_oracle_threads())
