"""Writing an aggregated Dataset to disk (output assembly).

The reference's scripts end with ``ds.to_netcdf(path)``.  xarray is not installed in this image,
so results returned as :class:`minixr.Dataset` are written with ``scipy.io.netcdf_file`` (NetCDF-3,
64-bit offset): the region dimension keeps its ``agglev`` labels (strings as fixed-width char
arrays, the CF convention xarray reads back as strings), ``time`` keeps the YYYYDDD integers
``tas_poly`` produced (transformations.py:195-199), variable attrs (units, description) are kept.
Real xarray Datasets are handed to their own ``to_netcdf``.  This is file IO around the path, not
part of the arithmetic.
"""
from __future__ import annotations

import numpy as np

__all__ = ["to_netcdf", "read_netcdf"]


def _label_array(values):
    """1-D labels -> (array writable by NetCDF-3, extra char dimension length or None)."""
    v = np.asarray(values)
    if v.dtype.kind in "OUS":
        enc = np.char.encode(v.astype(str), "utf-8") if v.dtype.kind != "S" else v
        width = max(1, enc.dtype.itemsize)
        return np.frombuffer(np.ascontiguousarray(enc.astype("S%d" % width)).tobytes(), dtype="S1").reshape(len(v), width), width
    if v.dtype.kind in "iu":
        if v.size and (v.max() > np.iinfo(np.int32).max or v.min() < np.iinfo(np.int32).min):
            return v.astype(np.float64), None                # NetCDF-3 has no 64-bit integers
        return v.astype(np.int32), None
    if v.dtype.kind == "M":                                  # datetime64 -> days since 1970-01-01
        return v.astype("datetime64[D]").astype(np.int64).astype(np.int32), None
    return v.astype(np.float64), None


def to_netcdf(ds, path):
    """Write ``ds`` (minixr or xarray Dataset) to ``path``."""
    if hasattr(ds, "to_netcdf"):
        return ds.to_netcdf(path)
    import scipy.io
    with scipy.io.netcdf_file(path, "w", version=2) as f:
        for d, n in ds.dims.items():
            f.createDimension(d, int(n))
        for name, c in ds.coords.items():
            arr, width = _label_array(c.values)
            dims = tuple(c.dims)
            if width is not None:
                f.createDimension(name + "_strlen", width)
                dims = dims + (name + "_strlen",)
            var = f.createVariable(name, arr.dtype.char if arr.dtype.kind != "S" else "S1", dims)
            var[:] = arr
            if np.asarray(c.values).dtype.kind == "M":
                var.units = "days since 1970-01-01"
        for name, v in ds.data_vars.items():
            vals = np.asarray(v.values)
            if vals.dtype not in (np.float32, np.float64):
                vals = vals.astype(np.float64)
            var = f.createVariable(name, vals.dtype.char, tuple(v.dims))
            var[:] = vals
            for k, a in getattr(v, "attrs", {}).items():
                setattr(var, k, a if isinstance(a, (int, float)) else str(a))
    return path


def read_netcdf(path):
    """Read a file written by :func:`to_netcdf` back into a minixr Dataset (round-trip checks)."""
    import scipy.io
    from . import minixr
    out = minixr.Dataset()
    with scipy.io.netcdf_file(path, "r", mmap=False) as f:
        names = list(f.variables)
        coords = [n for n in names if n in f.dimensions]
        for n in names:
            var = f.variables[n]
            vals = np.array(var[:])
            dims = tuple(var.dimensions)
            if dims and dims[-1] == n + "_strlen":
                vals = np.array([b"".join(row).decode("utf-8").rstrip("\x00 ") for row in vals], dtype=object)
                dims = dims[:-1]
            arr = minixr.DataArray(vals, dims, name=n)
            arr.attrs = {k: (a.decode() if isinstance(a, bytes) else a) for k, a in var._attributes.items()}
            if n in coords:
                out.coords[n] = arr
            else:
                out.data_vars[n] = arr
    return out
