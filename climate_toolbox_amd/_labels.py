"""Label work ahead of the plan, on the host in fp64: the exact-equality join of segment labels to grid labels
(aggregations.py:27, S1), the per-row backup fill (:73, S4) and the sorted-unique factorisation of region labels (:78, S3) --
each in native code behind include/wagg.h (wagg_resolve_cells / wagg_backup_fill / wagg_factorize_*), memoised per table
content (_memo.py).

Split out of aggregations.py in round 6; aggregations.py re-exports every name."""
from __future__ import annotations

import numpy as np
import pandas as pd

from ._memo import _f64, _frozen, _memo


def _native():
    from . import _lib
    return _lib, _lib.load()


def _resolve_cells(lat, lon, seg_lat, seg_lon, lon_major=False):
    """Exact-equality join of the segment labels to the grid labels in native code
    (``wagg_resolve_cells``; ``Dataset.sel`` without ``method=``, aggregations.py:27; S1).
    A missing label raises KeyError, like the reference.  Memoised per (grid labels, table columns)."""
    lat, lon, sa, so = _f64(lat), _f64(lon), _f64(seg_lat), _f64(seg_lon)
    return _memo("cells", (lat, lon, sa, so), lambda: _frozen(_resolve_cells_impl(lat, lon, sa, so, lon_major)),
                 extra=str(bool(lon_major)))


def _resolve_cells_impl(lat, lon, sa, so, lon_major):
    import ctypes as C
    _lib, L = _native()
    if sa.shape != so.shape or sa.ndim != 1:
        raise ValueError("segment lat/lon columns must be 1-D and of equal length")
    cell = np.empty(len(sa), dtype=np.int32)
    bad = C.c_int64(-1)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = L.wagg_resolve_cells(p(lat, C.c_double), len(lat), p(lon, C.c_double), len(lon), p(sa, C.c_double),
                              p(so, C.c_double), len(sa), 1 if lon_major else 0, p(cell, C.c_int32), C.byref(bad))
    if rc == _lib.EKEY:
        raise KeyError("not all values found in index: row %d (lat %r, lon %r)"
                       % (bad.value, float(sa[bad.value]), float(so[bad.value])))
    _lib.check(rc, "wagg_resolve_cells")
    return cell


def _exact_index(coord_values, wanted, name):
    """1-D form of the exact label lookup (kept for callers that need the two axes separately)."""
    idx = pd.Index(np.asarray(coord_values))
    pos = idx.get_indexer(np.asarray(wanted))
    if (pos < 0).any():
        bad = np.asarray(wanted)[pos < 0][:5]
        raise KeyError("not all values found in index %r: %r" % (name, bad.tolist()))
    return pos.astype(np.int64)


def _backup_fill(w, backup):
    """aggregations.py:73 per-row fill (``wagg_backup_fill``): w if w > 0 else backup (S4)."""
    import ctypes as C
    _lib, L = _native()
    w, backup = _f64(w), _f64(backup)
    out = np.empty_like(w)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    _lib.check(L.wagg_backup_fill(p(w), p(backup), len(w), p(out)), "wagg_backup_fill")
    return out


def _is_null_label(v):
    if v is None:
        return True
    try:
        return bool(v != v)
    except Exception:
        return False


def _factorize_labels(labels):
    """Sorted unique labels and per-row codes, -1 for null labels (xarray groupby, :78; S3), in
    native code for integer and string labels (``wagg_factorize_i64`` / ``_bytes``); other label
    types go through ``pandas.factorize(sort=True)``, which has the same contract.  Memoised per label
    column content (see _TABLE_MEMO); the codes come back read-only, the unique labels as a fresh copy."""
    labels = np.asarray(labels)

    def compute():
        uniq, codes = _factorize_labels_impl(labels)
        return np.array(uniq, copy=True), _frozen(codes)

    uniq, codes = _memo("factorize", (labels,), compute)
    return uniq.copy(), codes


def _factorize_labels_impl(labels):
    import ctypes as C
    n = len(labels)
    kind = labels.dtype.kind
    if kind in "iu" and labels.dtype.itemsize <= 8 and not (kind == "u" and labels.dtype.itemsize == 8):
        _lib, L = _native()
        lab = np.ascontiguousarray(labels, dtype=np.int64)
        codes = np.empty(n, dtype=np.int32)
        uniq = np.empty(n, dtype=np.int64)
        nu = C.c_int64(0)
        _lib.check(L.wagg_factorize_i64(lab.ctypes.data_as(C.POINTER(C.c_int64)), None, n,
                                        codes.ctypes.data_as(C.POINTER(C.c_int32)),
                                        uniq.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nu)), "wagg_factorize_i64")
        return uniq[:nu.value].astype(labels.dtype), codes
    if kind in "OUS" and n:
        null = None
        is_str = kind != "O"
        if kind == "O":
            what = pd.api.types.infer_dtype(labels, skipna=False)           # C loop, no Python objects made
            if what != "string" and pd.api.types.infer_dtype(labels, skipna=True) == "string":
                null, what = pd.isna(labels), "string"
            is_str = what == "string"
        if is_str:
            _lib, L = _native()
            filled = labels if null is None else np.where(null, "", labels)
            if kind == "S":
                enc = filled
            else:
                try:
                    enc = filled.astype("S")                                 # ASCII labels: one C pass
                except UnicodeEncodeError:
                    enc = np.char.encode(filled.astype(str), "utf-8")        # UTF-8 keeps code point order
            enc = np.ascontiguousarray(enc)
            width = enc.dtype.itemsize
            if width == 0:
                return np.array([], dtype=object), np.full(n, -1, dtype=np.int32)
            codes = np.empty(n, dtype=np.int32)
            rows = np.empty(n, dtype=np.int64)
            nu = C.c_int64(0)
            nm = None if null is None else np.ascontiguousarray(null, dtype=np.uint8)
            _lib.check(L.wagg_factorize_bytes(enc.ctypes.data_as(C.c_char_p), width,
                                              None if nm is None else nm.ctypes.data_as(C.POINTER(C.c_uint8)), n,
                                              codes.ctypes.data_as(C.POINTER(C.c_int32)),
                                              rows.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nu)),
                       "wagg_factorize_bytes")
            return labels[rows[:nu.value]], codes
    codes, uniq = pd.factorize(labels, sort=True)
    return np.asarray(uniq), np.asarray(codes, dtype=np.int32)
