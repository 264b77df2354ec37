"""Drop-in for the grid-level polynomial transform in front of the aggregation path
(SURVEY.md section 8f-3), fused into the aggregation kernels.

Reference (read as text): climate_toolbox/transformations/transformations.py
  tas_poly(ds, power, varname)   :160-208   (tas - 273.15) ** power, leap days removed, time -> YYYYDDD
  snyder_edd(tasmin, tasmax, e)  :7-93      Snyder exceedance degree days (nested xr.where, :75-87)
  snyder_gdd(tasmin, tasmax, lo, hi) :96-144   EDD(lo) - EDD(hi)
  ordinal(n)                     :211-214
  climate_toolbox/utils/utils.py:74-77   remove_leap_days
  climate_toolbox/utils/utils.py:10-20   convert_kelvin_to_celsius

The reference evaluates the power on the whole grid on the host and hands the new grid to the
aggregation.  Here ``tas_poly`` returns a dataset whose variable still points at the Kelvin
buffer and carries ``(offset, power)``; ``weighted_aggregate_grid_to_regions`` evaluates the
transform on the GPU while the data is loaded (``wagg_apply_poly_*``), so the transformed grid is
never written anywhere.  ``tas_poly_aggregate`` does several powers in ONE pass over the data.
"""
from __future__ import annotations

import numpy as np

from . import minixr
from . import aggregations as _agg

__all__ = ["tas_poly", "tas_poly_aggregate", "snyder_edd", "snyder_gdd", "ordinal", "remove_leap_days",
           "convert_kelvin_to_celsius"]

KELVIN = 273.15


def ordinal(n):
    """1 -> "1st", 2 -> "2nd", 3 -> "3rd", 4 -> "4th", 11 -> "11th", 22 -> "22nd" ...: the wording of the power in tas_poly's
    description attribute (same outputs as the reference's helper, transformations.py:211-214)."""
    n = int(n)
    last_two, last = n % 100, n % 10
    if 11 <= last_two <= 13 or last in (0, 4, 5, 6, 7, 8, 9):
        suffix = "th"
    else:
        suffix = {1: "st", 2: "nd", 3: "rd"}[last]
    return str(n) + suffix


def _month_day(time_values):
    t = np.asarray(time_values).astype("datetime64[D]")
    months = t.astype("datetime64[M]")
    month = months.astype(int) % 12 + 1
    day = (t - months.astype("datetime64[D]")).astype(int) + 1
    year = t.astype("datetime64[Y]").astype(int) + 1970
    return year, month, day


def remove_leap_days(ds):
    """Drop every 29 February along ``time`` (utils.py:74-77).  A dataset without one is returned as
    it is (no copy)."""
    _, month, day = _month_day(ds.coords["time"].values)
    keep = ~((month == 2) & (day == 29))
    if keep.all():
        return ds
    out = minixr.Dataset()
    for k, c in ds.coords.items():
        out.coords[k] = minixr.DataArray(np.asarray(c.values)[keep], c.dims) if c.dims == ("time",) else c
    for k, v in ds.data_vars.items():
        if "time" in v.dims:
            ax = v.dims.index("time")
            raw = _compress(v._values, keep, ax)
            if isinstance(v, minixr.LazyArray):            # keeps lon order, transform, degree-day partner, attrs
                edd_raw = None if v._edd is None else _compress(v._edd[0], keep, ax)
                out.data_vars[k] = v._replace(raw=raw, edd_raw=edd_raw, name=k)
            else:
                out.data_vars[k] = minixr.DataArray(raw, v.dims, name=k)
                out.data_vars[k].attrs = dict(getattr(v, "attrs", {}))
        else:
            out.data_vars[k] = v
    return out


def _compress(buf, keep, axis):
    if type(buf).__module__.startswith("torch"):          # device-resident buffer: wagg_take_axis
        from . import engine
        return engine.take_axis(buf, axis, np.flatnonzero(keep))
    return np.compress(keep, np.asarray(buf), axis=axis)


def convert_kelvin_to_celsius(df, temp_name):
    """Convert Kelvin to Celsius (utils.py:10-20) -- lazily: the variable keeps its Kelvin buffer
    and carries the offset, which the aggregation applies while loading."""
    v = df[temp_name]
    if getattr(v, "_xform", None) is not None or getattr(v, "_edd", None) is not None:
        raise ValueError("%r already carries a lazy transform" % (temp_name,))
    attrs = dict(getattr(v, "attrs", {}))
    attrs.update({"units": "C", "valid_min": -108.78788, "valid_max": 62.02828})
    df.data_vars[temp_name] = minixr.LazyArray(v._values, v.dims, lon_perm=getattr(v, "_lon_perm", None),
                                               xform=(-KELVIN, 1), name=temp_name, attrs=attrs)
    return df


def _day_index(ds):
    """transformations.py:191-199: ``time`` -> YYYYDDD integers (at most 365 days per call)."""
    ntime = len(ds.coords["time"].values)
    if ntime > 365:
        raise ValueError
    year, _, _ = _month_day(ds.coords["time"].values)
    return year * 1000 + np.arange(1, ntime + 1)


def _describe(power):
    raised = "" if power == 1 else " raised to the {powername} power".format(powername=ordinal(power))
    return ("Daily average temperature (degrees C){raised}\n\n"
            "            Leap years are removed before counting days (uses a 365 day\n"
            "            calendar).").format(raised=raised).strip()


def tas_poly(ds, power, varname):
    """Drop-in for transformations.py:160-208: ``(tas - 273.15) ** power`` as variable ``varname``,
    29 February dropped, ``time`` relabelled to YYYYDDD integers (at most 365 days per call).

    The returned variable is lazy (see the module docstring); ``.values`` evaluates it on demand.
    """
    if int(power) != power or power < 1:
        raise ValueError("power must be a positive integer, got %r" % (power,))
    power = int(power)
    description = _describe(power)
    ds = remove_leap_days(ds)
    tas = ds["tas"]
    if getattr(tas, "_xform", None) is not None:
        raise ValueError("'tas' already carries a lazy transform")
    day = _day_index(ds)
    ds1 = minixr.Dataset()
    for k, c in ds.coords.items():
        ds1.coords[k] = minixr.DataArray(day, ("time",)) if k == "time" else c
    attrs = {"units": "C^{}".format(power) if power > 1 else "C", "long_title": description.splitlines()[0],
             "description": description, "variable": varname}
    ds1.data_vars[varname] = minixr.LazyArray(tas._values, tas.dims, lon_perm=getattr(tas, "_lon_perm", None),
                                              xform=(-KELVIN, power), name=varname, attrs=attrs)
    return ds1


def tas_poly_aggregate(ds, powers, aggwt, agglev, weights, varnames=None, backup_aggwt="areawt"):
    """``tas_poly`` for several powers followed by ``weighted_aggregate_grid_to_regions`` of each --
    as ONE pass over the temperature field (powers 1..4 of fp32 (time, lat, lon) data share a single
    read of the grid from HBM).  ``varnames`` defaults to ``tas-poly-<p>``.  Returns one Dataset
    with a variable per power, dims/coords as the reference's aggregation gives them."""
    powers = [int(p) for p in powers]
    if not powers or min(powers) < 1 or len(set(powers)) != len(powers):
        raise ValueError("powers must be distinct positive integers, got %r" % (powers,))
    if varnames is None:
        varnames = ["tas-poly-%d" % p for p in powers]
    if len(varnames) != len(powers):
        raise ValueError("one variable name per power")
    if isinstance(weights, str):
        weights = _agg.prepare_spatial_weights_data(weights)
    ds = remove_leap_days(ds)
    day = _day_index(ds)
    ds = minixr.Dataset({"tas": ds["tas"]}, coords={k: (minixr.DataArray(day, ("time",)) if k == "time" else c)
                                                  for k, c in ds.coords.items()})
    re = _agg._reindex_spatial_data_to_regions(ds, weights)
    res, rdims, coords, was_xr = _agg._aggregate_core(re, "tas", aggwt, agglev, weights, backup_aggwt,
                                                      powers=powers, offset=-KELVIN)
    return _agg._as_dataset(dict(zip(varnames, res)), rdims, coords, was_xr)


def _units(arr):
    return getattr(arr, "attrs", {}).get("units")


def _degree_days(tasmin, tasmax, terms, units):
    for a in (tasmin, tasmax):
        if getattr(a, "_edd", None) is not None or (getattr(a, "_xform", None) is not None and a._xform[1] != 1):
            raise ValueError("degree days need plain (or Kelvin-shifted) temperature fields")
    off_lo = tasmin._xform[0] if getattr(tasmin, "_xform", None) is not None else 0.0
    off_hi = tasmax._xform[0] if getattr(tasmax, "_xform", None) is not None else 0.0
    if off_lo != off_hi:
        raise ValueError("tasmin and tasmax carry different offsets")
    lo, hi = tasmin._values, tasmax._values
    if tuple(tasmin.dims) != tuple(tasmax.dims) or lo.shape != hi.shape or lo.dtype != hi.dtype:
        raise ValueError("tasmin and tasmax must have the same dims, shape and dtype")
    plo, phi = getattr(tasmin, "_lon_perm", None), getattr(tasmax, "_lon_perm", None)
    if (plo is None) != (phi is None) or (plo is not None and not np.array_equal(plo, phi)):
        raise ValueError("tasmin and tasmax must share their longitude order")
    # check to make sure tasmax > tasmin everywhere (transformations.py:62), on the device
    # (wagg_any_less_*).  Host buffers are uploaded for this check and again by the aggregation:
    # hand device tensors in to avoid both copies.
    from . import engine
    engine.require_gpu()
    bad = engine.any_less(engine.to_device(hi), engine.to_device(lo))
    assert not bad, "values encountered where tasmin > tasmax"
    return minixr.LazyArray(lo, tasmin.dims, lon_perm=plo, edd=(hi, off_lo, terms), name=tasmin.name,
                            attrs={"units": units})


def snyder_edd(tasmin, tasmax, threshold):
    """Drop-in for transformations.py:7-93: degree days above ``threshold`` of the sinusoid through
    the daily (tasmin, tasmax) pair -- both DataArrays of one Dataset, in degrees C like the
    threshold.  The result is a lazy variable: assign it to a Dataset and hand that to
    ``weighted_aggregate_grid_to_regions`` -- the degree days are evaluated on the GPU while both
    fields are loaded (``wagg_apply_edd_*``); ``.values`` materialises the grid on demand.
    """
    assert _units(tasmin) == _units(tasmax)                    # :56, same units on both fields
    return _degree_days(tasmin, tasmax, [(1.0, float(threshold))],
                        "degreedays_{}{}".format(threshold, _units(tasmax)))


def snyder_gdd(tasmin, tasmax, threshold_low, threshold_high):
    """Drop-in for transformations.py:96-144: EDD(threshold_low) - EDD(threshold_high).  Lazy like
    :func:`snyder_edd`; the aggregation of the difference is the difference of the two aggregations.
    """
    assert _units(tasmin) == _units(tasmax)                    # :133
    return _degree_days(tasmin, tasmax, [(1.0, float(threshold_low)), (-1.0, float(threshold_high))],
                        "degreedays_{}-{}{}".format(threshold_low, threshold_high, _units(tasmax)))
