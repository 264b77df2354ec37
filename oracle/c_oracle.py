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
        for name, xp in (("wagg_oracle_segments_f32", C.POINTER(C.c_float)),
                         ("wagg_oracle_segments_f64", f64p)):
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
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def segments(X, cell_idx, region_code, w_eff, R, layout="TG"):
    """fp64 (T, R) result of the faithful single-threaded C restatement."""
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
    out = np.empty((T, R), dtype=np.float64)
    fn = L.wagg_oracle_segments_f32 if X.dtype == np.float32 else L.wagg_oracle_segments_f64
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


def threads():
    return int(lib().wagg_oracle_threads())
