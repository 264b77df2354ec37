"""What ``_reindex_spatial_data_to_regions`` returns (aggregations.py:8-32): the lazy stand-in for the gathered dataset -- the
source arrays plus the cell index of every segment row; nothing is copied unless ``.values`` is asked for -- and the
helpers that take a Dataset (xarray or minixr) apart.

Split out of aggregations.py in round 6; aggregations.py re-exports every name."""
from __future__ import annotations

import numpy as np

from . import engine as _engine, minixr
from ._layout import _flatten_for_device, _is_device_tensor, _result_dims, _spatial_layout, _to_device
from .engine import gather as _device_gather, require_gpu

try:  # optional: return real xarray objects when the caller hands us xarray objects
    import xarray as _xr
except Exception:  # pragma: no cover - xarray is absent from this image
    _xr = None


def _is_xarray(obj):
    return _xr is not None and isinstance(obj, (_xr.Dataset, _xr.DataArray))


class _ReindexedArray:
    """Lazy stand-in for one gathered data variable of aggregations.py:27."""

    def __init__(self, owner, name):
        self._owner, self.name = owner, name

    @property
    def dims(self):
        return _result_dims(self._owner._src_dims[self.name], "reshape_index")

    @property
    def shape(self):
        src_dims = self._owner._src_dims[self.name]
        src_shape = dict(zip(src_dims, self._owner._src_values[self.name].shape))
        return tuple(self._owner._nseg if d == "reshape_index" else src_shape[d] for d in self.dims)

    @property
    def dtype(self):
        return self._owner._src_values[self.name].dtype

    @property
    def values(self):
        return self._owner._materialise(self.name)

    def isnull(self):
        return minixr.DataArray(self.values, self.dims).isnull()

    def __array__(self, dtype=None, copy=None):
        v = self.values
        return v.astype(dtype) if dtype is not None else v


class ReindexedDataset(minixr.Dataset):
    """What ``_reindex_spatial_data_to_regions`` returns here: the source arrays plus the cell
    index of every segment row.  Nothing is copied until ``.values`` of a variable is asked for
    (then a device gather kernel runs); ``_aggregate_reindexed_data_to_regions`` never
    materialises it -- the gather is fused into the aggregation kernel."""

    def __init__(self, src_values, src_dims, coords, ilat, ilon, seg_lat, seg_lon, was_xarray,
                 lon_perms=None, xforms=None, edds=None):
        super().__init__()
        self._src_values, self._src_dims = src_values, src_dims
        self._ilat, self._ilon = ilat, ilon
        self._lon_perms = dict(lon_perms or {})     # variable -> file column of each sorted lon label
        self._xforms = dict(xforms or {})           # variable -> (offset, power), evaluated on the GPU
        self._edds = dict(edds or {})               # variable -> (tasmax buffer, offset, [(coef, threshold)])
        self._nseg = len(ilat)
        self._was_xarray = was_xarray
        for k, v in coords.items():
            if k not in ("lat", "lon"):
                self.coords[k] = v
        self.coords["lat"] = minixr.DataArray(np.asarray(seg_lat), ("reshape_index",))
        self.coords["lon"] = minixr.DataArray(np.asarray(seg_lon), ("reshape_index",))
        for name in src_values:
            self.data_vars[name] = _ReindexedArray(self, name)

    @property
    def dims(self):
        out = {}
        for name in self._src_values:
            arr = self.data_vars[name]
            if isinstance(arr, _ReindexedArray):
                out.update(zip(arr.dims, arr.shape))
        for c in self.coords.values():
            for d, n in zip(c.dims, c.shape):
                out.setdefault(d, n)
        return out

    def _cell_index(self, name):
        dims = self._src_dims[name]
        shape = dict(zip(dims, self._src_values[name].shape))
        ia, io, *_ = _spatial_layout(dims)
        perm = self._lon_perms.get(name)
        ready = getattr(self, "_row_major_cell", None)
        if ready is not None and perm is None and ia < io and (shape["lat"], shape["lon"]) == ready[1:]:
            return ready[0], shape["lat"] * shape["lon"]                          # a PreparedWeights' own (frozen) index
        ilon = self._ilon if perm is None else np.asarray(perm)[self._ilon]
        if ia < io:
            return (self._ilat * shape["lon"] + ilon).astype(np.int32), shape["lat"] * shape["lon"]
        return (ilon * shape["lat"] + self._ilat).astype(np.int32), shape["lat"] * shape["lon"]

    def _materialise(self, name):
        torch = require_gpu()
        X2, layout, others_shape, unflatten = _flatten_for_device(self._src_values[name], self._src_dims[name])
        cell, _ = self._cell_index(name)
        Xd = _to_device(X2)
        ci = torch.from_numpy(np.ascontiguousarray(cell)).cuda()
        out = _device_gather(Xd, ci, layout=layout, out_layout="TR" if layout == "TG" else "RT")
        if name in self._xforms:                     # wagg_transform_poly_*: the kernels' own device function
            off, pw = self._xforms[name]
            out = _engine.transform_poly(out, off, pw)
        if name in self._edds:
            hi, off, terms = self._edds[name]
            H2 = _flatten_for_device(hi, self._src_dims[name])[0]
            hig = _device_gather(_to_device(H2), ci, layout=layout, out_layout="TR" if layout == "TG" else "RT")
            out = _engine.transform_edd(out, hig, off, terms)
        return unflatten(out.cpu().numpy(), self._nseg)


def _extract(ds):
    """Pull (values, dims) of every data variable and the 1-D coords out of an xarray or minixr
    Dataset."""
    if _is_xarray(ds):
        src_values = {k: v.values for k, v in ds.data_vars.items()}
        src_dims = {k: tuple(v.dims) for k, v in ds.data_vars.items()}
        coords = {k: minixr.DataArray(v.values, tuple(v.dims)) for k, v in ds.coords.items()}
        return src_values, src_dims, coords, True
    # a lazily lon-sorted variable (standardize.py) hands over its file-order buffer; the column
    # permutation is folded into the cell index by _reindex_spatial_data_to_regions
    src_values = {k: (v._values if isinstance(v, minixr.LazyArray) or _is_device_tensor(v._values) else v.values)
                  for k, v in ds.data_vars.items()}
    src_dims = {k: tuple(v.dims) for k, v in ds.data_vars.items()}
    return src_values, src_dims, dict(ds.coords), False


def _lon_perms(ds):
    if _is_xarray(ds):
        return {}
    return {k: v._lon_perm for k, v in ds.data_vars.items() if getattr(v, "_lon_perm", None) is not None}


def _edds(ds):
    """variable -> (tasmax buffer, offset, [(coef, threshold)]) of a lazy degree-day variable."""
    if _is_xarray(ds):
        return {}
    return {k: v._edd for k, v in ds.data_vars.items() if getattr(v, "_edd", None) is not None}


def _xforms(ds):
    """variable -> (offset, power) of a lazily transformed variable (transformations.tas_poly)."""
    if _is_xarray(ds):
        return {}
    return {k: v._xform for k, v in ds.data_vars.items() if getattr(v, "_xform", None) is not None}
