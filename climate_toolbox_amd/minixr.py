"""A very small labelled-array container used when ``xarray`` is not importable.

It exists only so that the drop-in functions of ``climate_toolbox_amd.aggregations`` have an
object to take and return in an environment without xarray (this image has none).  It models the
handful of attributes the reference path touches (aggregations.py:24-27, :64-82 and
tests/test_climate_toolbox.py:51-64,109-135): ``Dataset(data_vars, coords)``, ``ds[name]``,
``ds.name``, ``ds.dims``, ``ds.coords``, ``DataArray.values/.dims/.shape/.isnull()/.any()``.
It is NOT an xarray re-implementation and carries no arithmetic.
"""
from __future__ import annotations

import numpy as np


class DataArray:
    def __init__(self, values, dims=None, coords=None, name=None):
        self._values = values if hasattr(values, "shape") else np.asarray(values)
        if dims is None:
            dims = tuple("dim_%d" % i for i in range(self._values.ndim))
        if isinstance(dims, str):
            dims = (dims,)
        self.dims = tuple(dims)                       # iterating a dict yields its keys (ref :65)
        if len(self.dims) != self._values.ndim:
            raise ValueError("dims %r do not match array of shape %r" % (self.dims, self._values.shape))
        self.coords = dict(coords or {})
        self.name = name
        self.attrs = {}

    @property
    def values(self):
        v = self._values
        if type(v).__module__.startswith("torch"):      # a device-resident buffer: copy out on demand
            return v.detach().cpu().numpy()
        return np.asarray(v)

    @property
    def shape(self):
        return tuple(self._values.shape)

    @property
    def dtype(self):
        return self._values.dtype

    @property
    def ndim(self):
        return self._values.ndim

    @property
    def sizes(self):
        return dict(zip(self.dims, self.shape))

    def __len__(self):
        return self.shape[0]

    def isnull(self):
        v = self.values
        if v.dtype.kind == "f":
            return DataArray(np.isnan(v), self.dims)
        if v.dtype == object:
            return DataArray(np.array([x is None or x != x for x in v.ravel()]).reshape(v.shape), self.dims)
        return DataArray(np.zeros(v.shape, dtype=bool), self.dims)

    def notnull(self):
        return DataArray(~self.isnull().values, self.dims)

    def any(self):
        return bool(np.any(self.values))

    def all(self):
        return bool(np.all(self.values))

    def __bool__(self):
        return bool(self.values)

    def __array__(self, dtype=None, copy=None):
        v = self.values
        return v.astype(dtype) if dtype is not None else v

    def __repr__(self):
        return "<minixr.DataArray %s %r %s>" % (self.name or "", dict(zip(self.dims, self.shape)), self.dtype)


class LazyArray(DataArray):
    """A data variable that keeps the buffer it was loaded with and carries, instead of a copy,
    (a) a re-ordering of its ``lon`` axis (element j along ``lon`` is column ``lon_perm[j]`` of
    the buffer; standardize.py) and/or (b) an element transform ``(x + offset) ** power``
    (transformations.py).  The aggregation folds (a) into the plan's cell index and evaluates (b)
    while the data is loaded on the GPU; ``.values`` materialises the array on demand."""

    def __init__(self, raw, dims, lon_perm=None, xform=None, name=None, attrs=None, edd=None):
        super().__init__(raw, dims, name=name)
        self._lon_perm = None if lon_perm is None else np.asarray(lon_perm, dtype=np.int64)
        self._xform = None if xform is None else (float(xform[0]), int(xform[1]))
        # (c) Snyder degree days of (tasmin = this buffer, tasmax = edd[0]) shifted by edd[1]:
        # sum of coef * EDD(threshold) over edd[2] = [(coef, threshold), ...]  (transformations.py)
        self._edd = None if edd is None else (edd[0], float(edd[1]), [(float(c), float(e)) for c, e in edd[2]])
        self.attrs = dict(attrs or {})

    @property
    def values(self):
        raw = np.asarray(self._values)
        if self._lon_perm is not None and "lon" in self.dims:
            raw = np.take(raw, self._lon_perm, axis=self.dims.index("lon"))
        if self._xform is not None:
            from .engine import require_gpu
            torch = require_gpu()                       # evaluated on the device, like the fused path
            off, pw = self._xform
            t = torch.from_numpy(np.ascontiguousarray(raw)).cuda()
            raw = ((t + off) ** pw).cpu().numpy()
        if self._edd is not None:
            from .engine import require_gpu
            torch = require_gpu()
            hi, off, terms = self._edd
            hi = np.asarray(hi)
            if self._lon_perm is not None and "lon" in self.dims:
                hi = np.take(hi, self._lon_perm, axis=self.dims.index("lon"))
            lo_t = torch.from_numpy(np.ascontiguousarray(raw)).cuda() + off
            hi_t = torch.from_numpy(np.ascontiguousarray(hi)).cuda() + off
            raw = sum(c * snyder_edd_device(torch, lo_t, hi_t, e) for c, e in terms).cpu().numpy()
        return raw


def snyder_edd_device(torch, tmin, tmax, e):
    """transformations.py:64-87 with torch ops on device tensors (materialised ``.values`` only; the
    aggregation evaluates the same formula inside its load stage, csrc/wagg_sparse.hip)."""
    mean, width = (tmax + tmin) / 2, (tmax - tmin) / 2
    theta = torch.asin((e - mean) / width)
    inner = torch.where(tmax > e, ((mean - e) * (np.pi / 2 - theta) + width * torch.cos(theta)) / np.pi,
                        torch.zeros_like(mean))
    return torch.where(tmin < e, inner, mean - e)


class _Coords(dict):
    def __init__(self, owner):
        super().__init__()
        self._owner = owner

    def __setitem__(self, key, value):
        super().__setitem__(key, self._owner._as_array(key, value))


class Dataset:
    def __init__(self, data_vars=None, coords=None):
        self.coords = _Coords(self)
        self.data_vars = {}
        for k, v in (coords or {}).items():
            self.coords[k] = v
        for k, v in (data_vars or {}).items():
            self[k] = v

    def _as_array(self, key, value):
        if isinstance(value, DataArray):
            value.name = key
            return value
        if isinstance(value, tuple) and len(value) >= 2 and not hasattr(value, "shape"):
            dims, vals = value[0], value[1]
            return DataArray(np.asarray(vals) if not hasattr(vals, "shape") else vals, dims, name=key)
        arr = np.asarray(value)
        return DataArray(arr, (key,) if arr.ndim == 1 else None, name=key)

    @property
    def dims(self):
        out = {}
        for arr in list(self.data_vars.values()) + list(self.coords.values()):
            for d, n in zip(arr.dims, arr.shape):
                out.setdefault(d, n)
        return out

    sizes = dims

    def __getitem__(self, key):
        if key in self.data_vars:
            return self.data_vars[key]
        if key in self.coords:
            return self.coords[key]
        raise KeyError(key)

    def __setitem__(self, key, value):
        self.data_vars[key] = self._as_array(key, value)

    def __contains__(self, key):
        return key in self.data_vars or key in self.coords

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __repr__(self):
        return "<minixr.Dataset dims=%r vars=%r coords=%r>" % (
            self.dims, list(self.data_vars), list(self.coords))
