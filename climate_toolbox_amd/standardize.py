"""Drop-in for the reference's coordinate standardisation in front of the aggregation path
 -- without the full-array permutation copy.

Reference (read as text):
  climate_toolbox/io/io.py:6-24            standardize_climate_data(ds)
  climate_toolbox/utils/utils.py:33-40     convert_lons_split(ds, lon_name)
  climate_toolbox/utils/utils.py:23-30     convert_lons_mono(ds, lon_name)
  climate_toolbox/utils/utils.py:43-57     rename_coords_to_lon_and_lat(ds)

The reference relabels the longitudes to -180..180 and then re-orders every data variable along
``lon`` (``ds.sel(lon=np.sort(...))``): a permutation copy of the whole array on the host.  Here
the returned dataset carries the SORTED labels but keeps each variable's buffer as it came from the
file, together with the column permutation; ``_reindex_spatial_data_to_regions`` folds that
permutation into the plan's cell index, so a raw 0..360 field goes to the GPU untouched.
``.values`` of a variable still gives the permuted array (same numbers as the reference), made on
demand.
"""
from __future__ import annotations

import numpy as np

from . import minixr

__all__ = ["standardize_climate_data", "convert_lons_split", "convert_lons_mono",
           "rename_coords_to_lon_and_lat"]


def LonSortedArray(raw, dims, perm, name=None, xform=None, attrs=None, edd=None):
    """A data variable whose ``lon`` axis is logically re-ordered: element j along ``lon`` is
    column ``perm[j]`` of the stored buffer (a :class:`minixr.LazyArray`)."""
    return minixr.LazyArray(raw, dims, lon_perm=perm, xform=xform, name=name, attrs=attrs, edd=edd)


def _is_xarray(ds):
    try:
        import xarray as xr
    except Exception:
        return False
    return isinstance(ds, (xr.Dataset, xr.DataArray))


def rename_coords_to_lon_and_lat(ds):
    """utils.py:43-57: latitude -> lat; longitude (or long) -> lon; drop ``z`` and squeeze."""
    if _is_xarray(ds):                                         # xarray present: its own rename/drop
        names = {"latitude": "lat", "longitude": "lon"}
        if "longitude" not in ds.coords:
            names["long"] = "lon"
        ds = ds.rename({k: v for k, v in names.items() if k in ds.coords})
        return ds.drop_vars("z").squeeze() if "z" in ds.coords else ds
    ren = {}
    if "latitude" in ds.coords:
        ren["latitude"] = "lat"
    if "longitude" in ds.coords:
        ren["longitude"] = "lon"
    elif "long" in ds.coords:
        ren["long"] = "lon"
    drop_z = "z" in ds.coords

    def fix(arr):
        dims = tuple(ren.get(d, d) for d in arr.dims)
        raw = arr._values
        squeeze = lambda b: b
        if drop_z:                                            # .drop('z').squeeze(): all size-1 dims go
            keep = [i for i, n in enumerate(raw.shape) if n != 1]
            squeeze = lambda b: b.reshape([b.shape[i] for i in keep])
            raw = squeeze(raw)
            dims = tuple(dims[i] for i in keep)
        if isinstance(arr, minixr.LazyArray):                 # lon order, transform, degree-day partner, attrs stay
            return arr._replace(raw=raw, dims=dims, edd_raw=None if arr._edd is None else squeeze(arr._edd[0]))
        out = minixr.DataArray(raw, dims)
        out.attrs = dict(getattr(arr, "attrs", {}))
        return out

    out = minixr.Dataset()
    for k, c in ds.coords.items():
        if drop_z and k == "z":
            continue
        out.coords[ren.get(k, k)] = fix(c)
    for k, v in ds.data_vars.items():
        out.data_vars[k] = fix(v)
        out.data_vars[k].name = k
    return out


def _relabel_sorted(ds, lon_name, relabel):
    if _is_xarray(ds):                                        # xarray present: do what the reference does
        ds[lon_name] = relabel(ds[lon_name])
        return ds.sel(**{lon_name: np.sort(ds[lon_name].values)})
    old = np.asarray(ds.coords[lon_name].values)
    new = relabel(old)
    perm = np.argsort(new, kind="stable")
    sorted_lab = new[perm]
    if len(sorted_lab) > 1 and (np.diff(sorted_lab) == 0).any():
        # ``ds.sel`` on a non-unique index does not return one column per label
        raise ValueError("longitudes are not unique after relabelling (%s)" % lon_name)
    out = minixr.Dataset()
    for k, c in ds.coords.items():
        out.coords[k] = minixr.DataArray(sorted_lab, (lon_name,)) if k == lon_name else c
    for k, v in ds.data_vars.items():
        if lon_name in v.dims:
            prev = getattr(v, "_lon_perm", None)
            p = perm if prev is None else np.asarray(prev)[perm]
            if lon_name == "lon":
                arr = LonSortedArray(v._values, v.dims, p, name=k, xform=getattr(v, "_xform", None),
                                     attrs=getattr(v, "attrs", None), edd=getattr(v, "_edd", None))
            else:                                             # a differently named axis: permute eagerly
                arr = minixr.DataArray(np.take(v.values, perm, axis=v.dims.index(lon_name)), v.dims, name=k)
                arr.attrs = dict(getattr(v, "attrs", {}))
        else:
            arr = v
        out.data_vars[k] = arr
    return out


def convert_lons_split(ds, lon_name="longitude"):
    """Drop-in for utils.py:33-40 (0..360 labels -> -180..180, ascending), lazily along ``lon``."""
    return _relabel_sorted(ds, lon_name, lambda v: (v + 180) % 360 - 180)


def convert_lons_mono(ds, lon_name="longitude"):
    """Drop-in for utils.py:23-30 (-180..180 labels -> 0..360, ascending), lazily along ``lon``."""
    return _relabel_sorted(ds, lon_name, lambda v: v % 360)


def standardize_climate_data(ds):
    """Drop-in for io/io.py:6-24: coordinate names become ``lat`` / ``lon`` and the longitudes run
    from -180 to 180 in ascending order.  No data is moved (see the module docstring)."""
    return convert_lons_split(rename_coords_to_lon_and_lat(ds), lon_name="lon")
