"""TESTS-ONLY STAND-IN for the ``xarray`` package -- NOT xarray, not shipped, not importable unless a test puts
``tests/stubs`` on ``sys.path`` on purpose.

xarray is not installed in this image, so the branches of ``climate_toolbox_amd`` that take and return
real xarray objects (``aggregations._extract`` / ``_as_dataset`` / ``_is_xarray``, ``standardize.py``'s xarray routes,
``output.to_netcdf``) had never met an object of that type.  This module exposes ``Dataset`` and ``DataArray`` with exactly the
attributes those branches touch -- ``data_vars``, ``coords``, ``dims``, ``.values``, ``__getitem__`` / ``__setitem__``, label
``sel`` along one dimension, ``rename``, ``drop_vars``, ``squeeze``, ``to_netcdf``, the three arithmetic operators of the
longitude relabelling -- so that a test can EXECUTE those lines and compare their results with the minixr route.  It says
nothing about real xarray's behaviour beyond the documented meaning of these calls; the day a real xarray is available the
same test runs against it by dropping the path injection."""
import numpy as np

__version__ = "0.0-stand-in"


class DataArray:
    def __init__(self, data, dims=None, coords=None, name=None, attrs=None):
        self.values = np.asarray(data)
        if dims is None:
            dims = tuple("dim_%d" % i for i in range(self.values.ndim))
        self.dims = (dims,) if isinstance(dims, str) else tuple(dims)
        assert len(self.dims) == self.values.ndim, (self.dims, self.values.shape)
        self.coords = dict(coords or {})
        self.name = name
        self.attrs = dict(attrs or {})

    shape = property(lambda self: self.values.shape)
    dtype = property(lambda self: self.values.dtype)
    ndim = property(lambda self: self.values.ndim)

    def _like(self, v):
        return DataArray(v, self.dims, self.coords, self.name, self.attrs)

    def __add__(self, o):
        return self._like(self.values + getattr(o, "values", o))

    def __sub__(self, o):
        return self._like(self.values - getattr(o, "values", o))

    def __mod__(self, o):
        return self._like(self.values % getattr(o, "values", o))

    def __len__(self):
        return len(self.values)


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.coords, self.data_vars, self.attrs = {}, {}, dict(attrs or {})
        for k, v in (coords or {}).items():
            self.coords[k] = self._as_array(v, (k,), k)
        for k, v in (data_vars or {}).items():
            self.data_vars[k] = self._as_array(v, None, k)

    @staticmethod
    def _as_array(v, default_dims, name):
        if isinstance(v, DataArray):
            return DataArray(v.values, v.dims, None, name, v.attrs)
        if isinstance(v, tuple):                        # (dims, values[, attrs])
            return DataArray(v[1], v[0], None, name, v[2] if len(v) > 2 else None)
        return DataArray(v, default_dims, None, name)

    @property
    def dims(self):
        out = {}
        for a in list(self.coords.values()) + list(self.data_vars.values()):
            out.update(zip(a.dims, a.shape))
        return out

    def __contains__(self, k):
        return k in self.data_vars or k in self.coords

    def __getitem__(self, k):
        return self.data_vars[k] if k in self.data_vars else self.coords[k]

    def __getattr__(self, k):
        d = self.__dict__
        if k in d.get("data_vars", {}):
            return d["data_vars"][k]
        if k in d.get("coords", {}):
            return d["coords"][k]
        raise AttributeError(k)

    def __setitem__(self, k, v):
        target = self.coords if (k in self.coords or any(k in a.dims for a in self.data_vars.values())) else self.data_vars
        target[k] = self._as_array(v, (k,), k)

    def _copy(self):
        out = Dataset(attrs=self.attrs)
        out.coords = dict(self.coords)
        out.data_vars = dict(self.data_vars)
        return out

    def rename(self, names):
        out = Dataset(attrs=self.attrs)
        fix = lambda a, k: DataArray(a.values, tuple(names.get(d, d) for d in a.dims), None, names.get(k, k), a.attrs)
        out.coords = {names.get(k, k): fix(a, k) for k, a in self.coords.items()}
        out.data_vars = {names.get(k, k): fix(a, k) for k, a in self.data_vars.items()}
        return out

    def drop_vars(self, name):
        out = self._copy()
        out.coords.pop(name, None)
        out.data_vars.pop(name, None)
        return out

    def squeeze(self):
        out = Dataset(attrs=self.attrs)
        sq = lambda a: DataArray(a.values.reshape([n for n in a.shape if n != 1]), tuple(d for d, n in zip(a.dims, a.shape) if n != 1),
                                 None, a.name, a.attrs)
        out.coords = {k: (sq(a) if a.ndim > 1 or a.shape != (1,) else a) for k, a in self.coords.items()}
        out.data_vars = {k: sq(a) for k, a in self.data_vars.items()}
        return out

    def sel(self, **indexers):
        """exact label selection along named dimensions (no ``method=``: a missing label is a KeyError)"""
        out = self._copy()
        for dim, labels in indexers.items():
            have = np.asarray(out.coords[dim].values)
            labels = np.atleast_1d(np.asarray(getattr(labels, "values", labels)))
            order = {v: i for i, v in enumerate(have.tolist())}
            try:
                idx = np.array([order[v] for v in labels.tolist()], dtype=np.int64)
            except KeyError as e:
                raise KeyError("label %r not found along %r" % (e.args[0], dim))
            take = lambda a: DataArray(np.take(a.values, idx, axis=a.dims.index(dim)), a.dims, None, a.name, a.attrs) if dim in a.dims else a
            out.coords = {k: take(a) for k, a in out.coords.items()}
            out.data_vars = {k: take(a) for k, a in out.data_vars.items()}
        return out

    def to_netcdf(self, path):
        """enough of a file for the caller to see that ITS to_netcdf was the one called (output.to_netcdf hands xarray objects
        to their own method)"""
        np.savez(path, **{("coord_" + k): a.values for k, a in self.coords.items()},
                 **{("var_" + k): a.values for k, a in self.data_vars.items()})
        return path
