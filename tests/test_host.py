"""Host-side logic of the drop-in (label lookup, backup fill inputs, factorisation, layout
bookkeeping, weights CSV preparation) -- everything that runs before the C-ABI.  CPU only."""
import os

import numpy as np
import pandas as pd
import pytest

from climate_toolbox_amd import aggregations as A, minixr
from climate_toolbox_amd._lib import WaggError


def test_exact_index_is_exact_and_raises_keyerror():
    lat = np.arange(-89.875, 90, 2)
    np.testing.assert_array_equal(A._exact_index(lat, [lat[3], lat[0]], "lat"), [3, 0])
    with pytest.raises(KeyError):
        A._exact_index(lat, [lat[3] + 1e-9], "lat")          # S1: no tolerance, no nearest
    with pytest.raises(KeyError):
        A._exact_index(lat, [np.nan], "lat")


def test_factorize_labels_sorted_unique_and_nulls():
    u, c = A._factorize_labels(np.array([3, 1, 3, 2]))
    assert list(u) == [1, 2, 3] and list(c) == [2, 0, 2, 1]
    u, c = A._factorize_labels(np.array(["b", None, "a", "b", np.nan], dtype=object))
    assert list(u) == ["a", "b"] and list(c) == [1, -1, 0, 1, -1]   # S3
    u, c = A._factorize_labels(np.array([2.0, np.nan, 1.0]))
    assert list(u) == [1.0, 2.0] and list(c) == [1, -1, 0]
    u, c = A._factorize_labels(np.array(["USA.10", "USA.2", "CAN.1"], dtype=object))
    assert list(u) == ["CAN.1", "USA.10", "USA.2"]                    # plain string sort


@pytest.mark.parametrize("dims,exp", [
    (("time", "lat", "lon"), ("time", "g")), (("lat", "lon", "time"), ("g", "time")),
    (("lon", "lat", "time"), ("g", "time")), (("ens", "time", "lat", "lon"), ("ens", "time", "g")),
    (("lat", "time", "lon"), ("g", "time")), (("time", "lat", "lon", "ens"), ("time", "g", "ens"))])
def test_result_dims_S10(dims, exp):
    assert A._result_dims(dims, "g") == exp


def test_result_dims_needs_lat_lon_names():
    with pytest.raises(KeyError):                                   # S2: names are literal
        A._result_dims(("time", "latitude", "longitude"), "g")


@pytest.mark.parametrize("dims", [("time", "lat", "lon"), ("lat", "lon", "time"), ("lon", "lat", "time"),
                                  ("ens", "time", "lat", "lon"), ("lat", "time", "lon"),
                                  ("time", "lat", "lon", "ens"), ("lat", "lon")])
def test_flatten_unflatten_roundtrip(dims):
    """The (T x G)/(G x T) flattening must address cell (ilat, ilon) where _cell_index says."""
    sizes = {"time": 3, "lat": 4, "lon": 5, "ens": 2}
    shape = tuple(sizes[d] for d in dims)
    vals = np.random.default_rng(0).random(shape)
    X2, layout, _, unflatten = A._flatten_for_device(vals, dims)
    ia, io = dims.index("lat"), dims.index("lon")
    ilat, ilon = np.array([1, 3, 0]), np.array([4, 0, 2])
    cell = ilat * sizes["lon"] + ilon if ia < io else ilon * sizes["lat"] + ilat
    picked = X2[:, cell] if layout == "TG" else X2[cell, :].T           # (T, 3)
    idx = [slice(None)] * len(dims)
    idx[ia], idx[io] = ilat, ilon
    expect = vals[tuple(idx)]                                           # numpy pointwise gather
    # numpy puts the broadcast axis first when the advanced indices are separated; normalise
    first = min(ia, io)
    adjacent = abs(ia - io) == 1
    expect = np.moveaxis(expect, first if adjacent else 0, -1).reshape(-1, 3)
    np.testing.assert_array_equal(picked, expect)
    res = picked if layout == "TG" else picked.T
    back = unflatten(np.ascontiguousarray(res), 3)
    rd = A._result_dims(dims, "g")
    assert back.shape == tuple(3 if d == "g" else sizes[d] for d in rd)
    np.testing.assert_array_equal(np.moveaxis(back, rd.index("g"), -1).reshape(-1, 3), expect)


def test_prepare_spatial_weights_data(tmp_path):
    p = tmp_path / "w.csv"
    pd.DataFrame({"pix_cent_x": [180.125, 10.125, 180.125], "pix_cent_y": [1.0, 2.0, 1.0],
                  "areawt": [1.0, 2.0, 1.0], "hierid": ["a", "b", "a"]}).to_csv(p, index=False)
    df = A.prepare_spatial_weights_data(str(p))
    assert list(df["lon"]) == [-179.875, 10.125, -179.875]            # aggregations.py:144
    assert "lat" in df.columns and "pix_cent_x" not in df.columns     # :150
    assert df.index.names == ["reshape_index"]                        # :148
    assert len(df) == 3                                               # :147 drop_duplicates is a no-op
    assert A.prepare_spatial_weights_data(str(p)) is df               # memoised (:127)


def test_weights_none_raises_typeerror_like_reference():
    ds = minixr.Dataset({"t": (("time", "lat", "lon"), np.zeros((1, 2, 2)))},
                        coords={"lat": [0.0, 1.0], "lon": [0.0, 1.0]})
    with pytest.raises(TypeError):                                    # aggregations.py:118-119 vs :128
        A.weighted_aggregate_grid_to_regions(ds, "t", "areawt", "ISO")


def test_reindex_is_lazy_and_shapes_match_reference_test(ref_fixture):
    fx, _ = ref_fixture
    ds = minixr.Dataset({"temperature": (["lat", "lon", "time"], fx["temp"])},
                        coords={"lon": fx["lon"], "lat": fx["lat"], "time": np.arange(10)})
    df = pd.DataFrame({"lat": fx["seg_lat"], "lon": fx["seg_lon"]})
    out = A._reindex_spatial_data_to_regions(ds, df)
    # tests/test_climate_toolbox.py:115-116
    assert out.temperature.shape == (len(out["lon"]), len(out["time"])) == (100, 10)
    assert "reshape_index" in out.dims
    assert out.temperature.dims == ("reshape_index", "time")
    bad = df.copy()
    bad.loc[0, "lat"] = 0.123
    with pytest.raises(KeyError):
        A._reindex_spatial_data_to_regions(ds, bad)


def test_no_cpu_fallback_without_gpu(ref_fixture):
    """On a machine without a GPU the product path must raise, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    fx, _ = ref_fixture
    ds = minixr.Dataset({"temperature": (["lat", "lon", "time"], fx["temp"])},
                        coords={"lon": fx["lon"], "lat": fx["lat"], "time": np.arange(10)})
    df = pd.DataFrame({"lat": fx["seg_lat"], "lon": fx["seg_lon"], "areawt": fx["areawt"],
                       "popwt": fx["popwt"], "ISO": fx["ISO"]})
    with pytest.raises(WaggError):
        A.weighted_aggregate_grid_to_regions(ds, "temperature", "popwt", "ISO", df)


def test_synth_workloads_are_wellformed():
    from climate_toolbox_amd import synth
    lat, lon, tas, df = synth.c1_workload(T=3)
    assert tas.shape == (3, 90, 180) and df["hierid"].nunique() == 100
    cell, codes, w_eff, uniq = synth.code_segments(df, lat, lon, "popwt", "hierid")
    assert len(uniq) == 100 and (w_eff > 0).all() and cell.max() < 90 * 180
    lat2, lon2, df2 = synth.realistic_segments(nlat=90, nlon=180, R=300, seed=3)
    assert df2["hierid"].nunique() == 300
    per_cell = df2.groupby(["lat", "lon"]).size()
    assert per_cell.max() <= 3 and per_cell.min() >= 1


def test_host_block_plan_covers_the_rows_and_deals_blocks_round_robin():
    """The row-block pipeline's block plan (wagg_host_block_plan; no GPU needed): whole quanta per block, ~256 MiB of
    X, every row in exactly one block, at least two blocks per device when there are that many quanta, and block i on
    device slot i mod n -- the map the multi-device form (wagg_apply_host_multi_*) deals by."""
    from climate_toolbox_amd import _lib
    for T, row_bytes, q, n in [(700, 259200 * 4, 64, 1), (700, 259200 * 4, 64, 2), (18250, 1036800 * 4, 64, 8),
                               (10950, 1036800 * 4, 368, 8), (5, 4000, 64, 3), (0, 4000, 64, 2), (129, 8, 64, 4),
                               (2282, 1036800 * 8, 64, 1), (1000, 100, 176, 2)]:
        B, nb = _lib.host_block_plan(T, row_bytes, q, n)
        if T == 0:
            assert nb == 0
            continue
        assert nb == -(-T // B) and 1 <= B <= T
        assert B % q == 0 or B == T or B < q                       # whole quanta (a field shorter than one is one block)
        assert B * row_bytes <= max(256 << 20, q * row_bytes) or B <= q
        if T >= 2 * n * q:
            assert nb >= 2 * n                                     # copies and kernels of every device can overlap
        rows_of = [0] * n
        for i in range(nb):
            rows_of[i % n] += min(B, T - i * B)
        assert sum(rows_of) == T
        if nb >= n:
            assert max(rows_of) - min(rows_of) <= 2 * B           # round-robin keeps the devices within two blocks
    with pytest.raises(_lib.WaggError):
        _lib.host_block_plan(10, 0, 64, 1)


def test_dense_form_choice():
    """Structure-driven choice between the gather form and the dense MFMA form (host logic only)."""
    G, R, GB = 1036800, 24378, 1 << 30
    assert not A._prefer_dense(1.4 * G, G, R, True, "TG", 250 * GB)        # c2-real: compact regions
    assert A._prefer_dense(240 * G, G, R, True, "TG", 250 * GB)            # c5 uniform-random columns
    assert not A._prefer_dense(240 * G, G, R, False, "TG", 250 * GB)       # fp64: 202 GB of W would not fit
    # ... but a table filled to 1 % never becomes the full matrix (entry lists or fewer than half of the tiles)
    assert A._prefer_dense(240 * G, G, R, False, "TG", 250 * GB, nseg=int(0.01 * G * R))
    assert not A._prefer_dense(240 * G, G, R, False, "TG", 250 * GB, nseg=int(0.2 * G * R))
    assert A._prefer_dense(240 * G, G, 4000, False, "TG", 250 * GB)        # ... a smaller fp64 problem takes the f64 MFMA form
    assert not A._prefer_dense(240 * G, G, R, True, "GT", 250 * GB)        # fixture layout stays too
    assert not A._prefer_dense(240 * G, G, R, True, "TG", 100 * GB)        # 101 GB of W must fit


# ---------------------------------------------------------------------------------------------
# the label work in native code, checked against pandas / the oracle (no GPU needed)
# ---------------------------------------------------------------------------------------------
def test_native_resolve_cells_matches_oracle_and_raises():
    from oracle import ref_numpy as O
    lat, lon = np.arange(-89.875, 90, 2), np.arange(0.125, 360.0, 2)
    rng = np.random.default_rng(0)
    sl, so = rng.choice(lat, 500), rng.choice(lon, 500)
    cell = A._resolve_cells(lat, lon, sl, so)
    ref = O._lookup_dict(lat, sl, "lat") * len(lon) + O._lookup_dict(lon, so, "lon")
    np.testing.assert_array_equal(cell, ref)
    np.testing.assert_array_equal(A._resolve_cells(lat, lon, sl, so, lon_major=True),
                                  O._lookup_dict(lon, so, "lon") * len(lat) + O._lookup_dict(lat, sl, "lat"))
    for bad in (sl[3] + 1e-9, np.nan, 1e9):
        s2 = sl.copy(); s2[3] = bad
        with pytest.raises(KeyError, match="row 3"):
            A._resolve_cells(lat, lon, s2, so)                       # S1: exact, no nearest
    np.testing.assert_array_equal(A._resolve_cells(np.array([-0.0, 1.0]), np.array([5.0]), [0.0], [5.0]), [0])
    with pytest.raises(Exception, match="not unique"):
        A._resolve_cells(np.array([1.0, 1.0]), np.array([5.0]), [1.0], [5.0])


def test_native_backup_fill_and_relabel():
    from oracle import ref_numpy as O
    w = np.array([2.0, np.nan, 0.0, -1.0, 3.0, np.inf])
    b = np.array([1.0, 3.0, 2.0, 1.0, 9.0, np.nan])
    np.testing.assert_array_equal(A._backup_fill(w, b), O.effective_weights(w, b))
    np.testing.assert_array_equal(A._backup_fill(w, b), [2.0, 3.0, 2.0, 1.0, 3.0, np.inf])


@pytest.mark.parametrize("labels", [
    np.array([3, 1, 3, 2, -7]), np.array([3, 1, 3], dtype=np.int32), np.array(["b", "a", "b", "USA.10", "USA.2"]),
    np.array(["b", None, "a", "b", np.nan, "é", "z"], dtype=object), np.array([2.0, np.nan, 1.0]),
    np.array([], dtype=np.int64), np.array([None, None], dtype=object)])
def test_native_factorize_matches_pandas(labels):
    uniq, codes = A._factorize_labels(labels)
    c_ref, u_ref = pd.factorize(labels, sort=True)
    assert list(uniq) == list(u_ref)
    np.testing.assert_array_equal(codes, c_ref)
    assert codes.dtype == np.int32


# ---------------------------------------------------------------------------------------------
# standardize_climate_data folded into the plan (no permutation copy)
# ---------------------------------------------------------------------------------------------
def _raw_0_360_dataset(T=3, step=2.0, seed=0, names=("latitude", "longitude")):
    from climate_toolbox_amd import minixr
    rng = np.random.default_rng(seed)
    lat = np.arange(-89.0, 90, step)
    lon = np.arange(0.0, 360.0, step)                    # file order: 0 .. 358
    tas = rng.standard_normal((T, len(lat), len(lon)))
    ds = minixr.Dataset({"tas": (("time", names[0], names[1]), tas)},
                        coords={"time": np.arange(T), names[0]: lat, names[1]: lon})
    return ds, tas, lat, lon


def test_standardize_is_lazy_and_matches_reference_semantics():
    from climate_toolbox_amd.standardize import standardize_climate_data
    from climate_toolbox_amd.minixr import LazyArray
    from oracle import ref_numpy as O
    ds, tas, lat, lon = _raw_0_360_dataset()
    out = standardize_climate_data(ds)
    ref_vals, ref_lon = O.convert_lons_split(tas, ("time", "lat", "lon"), lon)
    assert set(out.coords) == {"time", "lat", "lon"} and out.tas.dims == ("time", "lat", "lon")
    np.testing.assert_array_equal(out.lon.values, ref_lon)
    assert (np.diff(out.lon.values) > 0).all() and out.lon.values.min() == -180.0
    assert isinstance(out.tas, LazyArray) and out.tas._values is ds.tas._values   # no copy
    np.testing.assert_array_equal(out.tas.values, ref_vals)                             # on demand


def test_standardize_permutation_is_folded_into_the_cell_index():
    from climate_toolbox_amd.standardize import standardize_climate_data
    ds, tas, lat, lon = _raw_0_360_dataset(names=("lat", "long"))
    out = standardize_climate_data(ds)
    rng = np.random.default_rng(1)
    seg_lat, seg_lon = rng.choice(lat, 50), rng.choice(out.lon.values, 50)   # -180..180 labels
    df = pd.DataFrame({"lat": seg_lat, "lon": seg_lon})
    re = A._reindex_spatial_data_to_regions(out, df)
    cell, G = re._cell_index("tas")
    assert G == len(lat) * len(lon)
    raw = ds.tas._values.reshape(3, -1)
    # the value picked from the FILE-order buffer is the value at that (lat, lon) label
    ilat = np.searchsorted(lat, seg_lat)
    raw_col = np.array([np.flatnonzero(((lon + 180) % 360 - 180) == v)[0] for v in seg_lon])
    np.testing.assert_array_equal(raw[:, cell], tas[:, ilat, raw_col])
    with pytest.raises(KeyError):
        A._reindex_spatial_data_to_regions(out, pd.DataFrame({"lat": [lat[0]], "lon": [200.0]}))


def test_standardize_drops_z_and_rejects_duplicate_longitudes():
    from climate_toolbox_amd import minixr
    from climate_toolbox_amd.standardize import standardize_climate_data, convert_lons_mono
    lat, lon = np.array([0.0, 1.0]), np.array([0.0, 90.0, 180.0, 270.0])
    v = np.arange(8.0).reshape(1, 2, 4)
    ds = minixr.Dataset({"tas": (("z", "lat", "lon"), v)}, coords={"z": np.array([5.0]), "lat": lat, "lon": lon})
    out = standardize_climate_data(ds)
    assert "z" not in out.coords and out.tas.dims == ("lat", "lon")
    np.testing.assert_array_equal(out.lon.values, [-180.0, -90.0, 0.0, 90.0])
    np.testing.assert_array_equal(out.tas.values, v[0][:, [2, 3, 0, 1]])
    back = convert_lons_mono(out, lon_name="lon")                     # and back to 0..360
    np.testing.assert_array_equal(back.lon.values, lon)
    np.testing.assert_array_equal(back.tas.values, v[0])
    bad = minixr.Dataset({"tas": (("lat", "lon"), v[0][:, :2])}, coords={"lat": lat, "lon": np.array([0.0, 360.0])})
    with pytest.raises(ValueError):
        standardize_climate_data(bad)


# ---------------------------------------------------------------------------------------------
# tas_poly as a lazy variable (host-side contract; the arithmetic is GPU-only)
# ---------------------------------------------------------------------------------------------
def _tas_ds(year=2001, T=365, nlat=6, nlon=8, dtype=np.float32, seed=0):
    from climate_toolbox_amd import minixr
    rng = np.random.default_rng(seed)
    time = np.arange("%d-01-01" % year, "%d-01-01" % (year + 1), dtype="datetime64[D]")[:T]
    lat, lon = np.linspace(-50, 50, nlat), np.linspace(-100, 100, nlon)
    tas = (288.15 + 8 * rng.standard_normal((len(time), nlat, nlon))).astype(dtype)
    return minixr.Dataset({"tas": (("time", "lat", "lon"), tas)}, coords={"time": time, "lat": lat, "lon": lon}), tas


def test_tas_poly_is_lazy_and_relabels_time_like_the_reference():
    from climate_toolbox_amd import tas_poly
    from climate_toolbox_amd.transformations import ordinal
    from climate_toolbox_amd.minixr import LazyArray
    ds, tas = _tas_ds()
    out = tas_poly(ds, 3, "tas-poly-3")
    v = out["tas-poly-3"]
    assert isinstance(v, LazyArray) and v._values is ds.tas._values and v._xform == (-273.15, 3)
    assert v.dims == ("time", "lat", "lon") and list(out.data_vars) == ["tas-poly-3"]
    np.testing.assert_array_equal(out.time.values, 2001000 + np.arange(1, 366))     # transformations.py:195
    assert v.attrs["units"] == "C^3" and "3rd power" in v.attrs["description"]
    assert tas_poly(ds, 1, "t").t.attrs["units"] == "C"
    got = [ordinal(n) for n in (1, 2, 3, 4, 11, 12, 13, 21, 22, 101)]
    assert got == ["1st", "2nd", "3rd", "4th", "11th", "12th", "13th", "21st", "22nd", "101st"]


def test_tas_poly_removes_leap_day_and_rejects_long_years():
    from climate_toolbox_amd import tas_poly
    ds, tas = _tas_ds(year=2000, T=366)                                             # leap year
    out = tas_poly(ds, 2, "p2")
    assert out.p2.shape[0] == 365
    np.testing.assert_array_equal(out.p2._values, np.delete(tas, 59, axis=0))       # 29 Feb = day 60
    np.testing.assert_array_equal(out.time.values, 2000000 + np.arange(1, 366))
    from climate_toolbox_amd import minixr
    time = np.arange("2001-01-01", "2002-02-01", dtype="datetime64[D]")
    long_ds = minixr.Dataset({"tas": (("time", "lat", "lon"), np.zeros((len(time), 1, 1), np.float32))},
                             coords={"time": time, "lat": np.zeros(1), "lon": np.zeros(1)})
    with pytest.raises(ValueError):
        tas_poly(long_ds, 2, "p2")


def test_to_netcdf_round_trip(tmp_path):
    """region time series with string region labels, YYYYDDD time and attrs survive a
    NetCDF-3 round trip."""
    from climate_toolbox_amd import minixr
    from climate_toolbox_amd.output import read_netcdf, to_netcdf
    rng = np.random.default_rng(0)
    labels = np.array(["USA.1.2", "CAN.10", "é-région"], dtype=object)
    vals = rng.standard_normal((4, 3)).astype(np.float32)
    ds = minixr.Dataset({"tas-poly-2": (("time", "hierid"), vals)},
                        coords={"time": 2001000 + np.arange(1, 5), "hierid": labels})
    ds["tas-poly-2"].attrs = {"units": "C^2", "variable": "tas-poly-2"}
    path = to_netcdf(ds, str(tmp_path / "out.nc"))
    back = read_netcdf(path)
    np.testing.assert_array_equal(back["tas-poly-2"].values, vals)
    assert back["tas-poly-2"].dims == ("time", "hierid") and back["tas-poly-2"].attrs["units"] == "C^2"
    np.testing.assert_array_equal(back.time.values, 2001000 + np.arange(1, 5))
    assert list(back.hierid.values) == list(labels)


def test_convert_lons_on_the_vectors_the_reference_tests_hold(golden_dir):
    """tests/test_climate_toolbox.py:180-197 through the drop-in functions (host logic only)."""
    from climate_toolbox_amd import convert_lons_mono, convert_lons_split, minixr
    h = np.load(os.path.join(golden_dir, "reference_held.npz"), allow_pickle=False)
    ds = convert_lons_mono(minixr.Dataset(coords={"lon": h["mono_lon"]}), lon_name="lon")
    np.testing.assert_array_equal(ds.lon.values, h["mono_expect"])
    ds = convert_lons_split(minixr.Dataset(coords={"longitude": h["split_lon"]}))
    np.testing.assert_array_equal(ds.longitude.values, h["split_expect"])


def test_table_memo_follows_the_table_content(monkeypatch):
    """The per-table memo of the label work: a repeated call with the same columns does no
    O(nseg) work, any in-place edit of a column is seen (object columns are fingerprinted by their pointer
    table with the hashed objects kept alive), and what comes back cannot poison the cache."""
    from climate_toolbox_amd import aggregations as A
    A._TABLE_MEMO.clear()
    rng = np.random.default_rng(5)
    n = 20000
    labels = np.array(["R%05d" % k for k in rng.integers(0, 900, n)], dtype=object)
    calls = {"n": 0}
    real = A._factorize_labels_impl

    def counting(lab):
        calls["n"] += 1
        return real(lab)

    monkeypatch.setattr(A._labels, "_factorize_labels_impl", counting)

    def check():
        uniq, codes = A._factorize_labels(labels)
        ec, eu = pd.factorize(labels, sort=True)
        np.testing.assert_array_equal(np.asarray(uniq, dtype=object), np.asarray(eu, dtype=object))
        np.testing.assert_array_equal(codes, ec.astype(np.int32))
        return uniq, codes

    uniq, codes = check()
    assert calls["n"] == 1
    uniq2, codes2 = check()
    assert calls["n"] == 1                                   # memo hit
    assert uniq2 is not uniq and not codes2.flags.writeable
    uniq2[0] = "poison"                                      # the caller's copy, not the cache's
    with pytest.raises(ValueError):
        codes2[0] = 7
    check()
    assert calls["n"] == 1
    # in-place edits, including a chain of replacements in one slot (freed strings may hand their address on)
    for k, new in enumerate(["ZZZ_%d" % i for i in range(6)]):
        labels[17] = new
        check()
        assert calls["n"] == 2 + k
    labels[17] = "R00001"
    check()
    # the exact-match label join: memoised per (grid, table columns), redone after an in-place edit
    lat, lon = np.arange(-44.5, 45, 1.0), np.arange(-89.5, 90, 1.0)
    sa, so = rng.choice(lat, n), rng.choice(lon, n)
    c1 = A._resolve_cells(lat, lon, sa, so)
    assert A._resolve_cells(lat, lon, sa, so) is c1 and not c1.flags.writeable
    so[5] = lon[0] if so[5] != lon[0] else lon[1]
    c2 = A._resolve_cells(lat, lon, sa, so)
    assert c2 is not c1 and c2[5] % len(lon) == (0 if so[5] == lon[0] else 1)
    sa[9] = 0.123                                            # not on the grid: KeyError, and nothing is cached
    with pytest.raises(KeyError):
        A._resolve_cells(lat, lon, sa, so)
    with pytest.raises(KeyError):
        A._resolve_cells(lat, lon, sa, so)
    assert len(A._TABLE_MEMO) <= A._TABLE_MEMO_MAX


def test_table_memo_label_types_without_a_buffer_or_a_safe_identity(monkeypatch):
    """datetime64 / timedelta64 label columns refuse the buffer protocol: they are fingerprinted through their
    int64 view and factorised as before.  Object columns whose elements are not str / bytes / None
    are never fingerprinted by pointer (equal pointers to a MUTABLE object do not mean equal labels): they are
    factorised afresh on every call."""
    from climate_toolbox_amd import aggregations as A
    A._TABLE_MEMO.clear()
    days = np.array(["2001-01-03", "2000-05-01", "2001-01-03", "1999-12-31"], dtype="datetime64[D]")
    for lab in (days, days - days[3]):
        uniq, codes = A._factorize_labels(lab)
        ec, eu = pd.factorize(lab, sort=True)
        np.testing.assert_array_equal(uniq, np.asarray(eu))
        np.testing.assert_array_equal(codes, ec.astype(np.int32))
        n = len(A._TABLE_MEMO)
        A._factorize_labels(lab)
        assert len(A._TABLE_MEMO) == n                       # a hit, not a second entry
    calls = {"n": 0}
    real = A._factorize_labels_impl

    def counting(lab):
        calls["n"] += 1
        return real(lab)

    monkeypatch.setattr(A._labels, "_factorize_labels_impl", counting)
    n = len(A._TABLE_MEMO)
    for lab in (np.array([3, "a", 2.5, "a"], dtype=object), np.array([(1, 2), (0, 1), (1, 2)] + [None], dtype=object)[:3]):
        with pytest.raises(A._Unhashable):
            A._raw_view(lab)
    mixed = np.array([7, 3, 7, 5], dtype=object)             # Python ints as objects: "integer", not pointer-safe
    for k in range(2):
        uniq, codes = A._factorize_labels(mixed)
        np.testing.assert_array_equal(codes, [2, 0, 2, 1])
        assert calls["n"] == k + 1 and len(A._TABLE_MEMO) == n
    nulls = np.array(["b", None, "a", np.nan, "b"], dtype=object)      # str + nulls stays memoised
    A._factorize_labels(nulls)
    A._factorize_labels(nulls)
    assert calls["n"] == 3 and len(A._TABLE_MEMO) == n + 1


def test_plan_cache_leases_and_eviction_budget(monkeypatch):
    """The plan cache (host logic with stand-in plans): eviction passes over a plan that another thread has
    leased, a sparse-form miss does not charge the dense byte budget (a 100 GB dense plan was evicted
    for a table that then took the segment-table form), and concurrent calls on the module caches are serialised."""
    import threading
    from climate_toolbox_amd import aggregations as A

    class FakePlan:
        def __init__(self, nbytes):
            self._lease, self.closed, self.info = threading.Lock(), False, {"nnz": nbytes // 16}

        def close(self):
            self.closed = True

    GB = 1 << 30
    cache = A._PLAN_CACHE
    saved = dict(cache)
    cache.clear()
    try:
        a, b, c = FakePlan(100 * GB), FakePlan(1 * GB), FakePlan(1 * GB)
        cache["a"], cache["b"], cache["c"] = a, b, c
        with A._CACHE_LOCK:
            assert A._evict_plans(144 * GB, keep=7) == 0                  # under both limits: nothing goes
            assert not (a.closed or b.closed or c.closed)
            a._lease.acquire()                                            # "another thread is applying a"
            freed = A._evict_plans(50 * GB, keep=7)                       # over the byte budget: a is the oldest...
            assert not a.closed and "a" in cache                          # ... but leased: passed over
            assert b.closed and c.closed and freed == 2 * GB              # (the others go: still over budget)
            a._lease.release()
            assert A._evict_plans(50 * GB, keep=7) == 100 * GB and a.closed and not cache
        # count limit
        ps = [FakePlan(16) for _ in range(10)]
        for i, q in enumerate(ps):
            cache[i] = q
        with A._CACHE_LOCK:
            A._evict_plans(1 << 40, keep=7)
        assert [q.closed for q in ps] == [True] * 3 + [False] * 7
        # _drop_plan forgets exactly the failed plan
        A._drop_plan(ps[5])
        assert ps[5].closed and 5 not in cache and len(cache) == 6
    finally:
        cache.clear()
        cache.update(saved)
    # the memo under threads: same answers, no exception, bounded size
    A._TABLE_MEMO.clear()
    labs = [np.array(["R%03d" % ((i * 7 + k) % 50) for i in range(2000)], dtype=object) for k in range(6)]
    errs = []

    def work(k):
        try:
            for _ in range(20):
                for lab in labs[k % 3:k % 3 + 3]:
                    uniq, codes = A._factorize_labels(lab)
                    assert list(uniq[codes[:5]]) == list(lab[:5])
        except Exception as e:                                            # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs and len(A._TABLE_MEMO) <= A._TABLE_MEMO_MAX


def test_prepared_weights_is_a_snapshot_of_the_coded_table():
    """prepare_weights: backup fill, factorisation and the label join happen once; the object answers
    for the columns the reference reads, refuses a call for another (aggwt, agglev), resolves a grid once, raises KeyError
    for a label that is not on the grid (S1) and is not reached by later edits of the DataFrame.  Host-side only."""
    from climate_toolbox_amd import aggregations as A, prepare_weights
    rng = np.random.default_rng(9)
    lat, lon = np.arange(-44.5, 45, 1.0), np.arange(-89.5, 90, 1.0)
    n = 5000
    df = pd.DataFrame({"lat": rng.choice(lat, n), "lon": rng.choice(lon, n), "areawt": rng.uniform(0.1, 1, n),
                       "popwt": rng.uniform(-0.2, 1, n), "hierid": ["R%03d" % k for k in rng.integers(0, 70, n)]})
    df.loc[::9, "popwt"] = np.nan
    prep = prepare_weights(df, "popwt", "hierid", lat=lat, lon=lon)
    w = np.where(df["popwt"].values > 0, df["popwt"].values, df["areawt"].values)
    np.testing.assert_array_equal(prep["popwt"].values, w)                       # aggregations.py:73 per row
    ec, eu = pd.factorize(df["hierid"].values, sort=True)
    np.testing.assert_array_equal(prep.codes, ec.astype(np.int32))
    assert list(prep.uniq) == list(eu) and len(prep) == n
    cell, ilat, ilon = prep.cells_for(lat, lon)
    np.testing.assert_array_equal(lat[ilat], df["lat"].values)
    np.testing.assert_array_equal(lon[ilon], df["lon"].values)
    assert prep.cells_for(lat.copy(), lon.copy())[0] is cell                     # the same grid again: nothing recomputed
    assert prep.cells_for(lat[::-1].copy(), lon)[0] is not cell                  # another grid: its own entry
    assert prep.plan_key(cell, len(lat) * len(lon), 70, len(lon), True, "TG") == prep.plan_key(cell, len(lat) * len(lon), 70, len(lon), True, "TG")
    assert prep.plan_key(cell, len(lat) * len(lon), 70, len(lon), True, "TG") != prep.plan_key(cell, len(lat) * len(lon), 70, len(lon), False, "TG")
    with pytest.raises(ValueError, match="prepared for aggwt='popwt'"):
        prep.check("areawt", "hierid")
    with pytest.raises(KeyError):
        prep["cropwt"]
    with pytest.raises(KeyError, match="not all values found"):
        prep.cells_for(lat + 0.25, lon)
    df.loc[0, "popwt"] = 123.0                                                  # a snapshot: the edit does not reach it
    assert prep["popwt"].values[0] == w[0]
    with pytest.raises(ValueError):
        prep.w_eff[0] = 1.0                                                      # and nobody can edit the snapshot
    assert prepare_weights(prep, "popwt", "hierid") is prep


def test_host_devices_default_and_opt_in(monkeypatch):
    """host-resident fields stay on the caller's device unless HOST_DEVICES opts in; "all" is ignored
    inside a one-process-per-GPU job (every rank sees every device there); the row blocks are sized by the plan's own
    launch quantum.  Host-side logic only."""
    import torch
    from climate_toolbox_amd import aggregations as A
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(A, "HOST_DEVICES", None)
    assert A._host_devices() == []
    monkeypatch.setattr(A, "HOST_DEVICES", "all")
    assert A._host_devices() == list(range(8))
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert A._host_devices() == []                                   # torchrun rank: the other devices are not ours
    monkeypatch.setattr(A, "HOST_DEVICES", [2, 2, 5])
    assert A._host_devices() == [2, 2, 5]                            # an explicit list is taken as it is
    monkeypatch.setattr(A, "HOST_DEVICES", "every")
    with pytest.raises(ValueError):
        A._host_devices()

    class FakeSparse:
        pass

    class FakeDense(A.DensePlan):
        def __init__(self, form, dtype):                             # (no device object behind it)
            self.info, self.dtype = {"form": form}, dtype

        def close(self):
            pass

        __del__ = close

    assert A._host_quantum(FakeSparse()) == 64
    assert A._host_quantum(FakeDense(0, "float32")) == 368 and A._host_quantum(FakeDense(1, "float64")) == 176
    assert A._host_quantum(FakeDense(2, "float32")) == 128 and A._host_quantum(FakeDense(2, "float64")) == 64


_XARRAY_HOST_CHILD = r'''
import numpy as np
import xarray as xr
assert xr.__version__ == "0.0-stand-in"
from climate_toolbox_amd import aggregations as A, minixr, standardize_climate_data, convert_lons_mono
from climate_toolbox_amd.standardize import rename_coords_to_lon_and_lat
rng = np.random.default_rng(1)
lat, lon0 = np.arange(-9.5, 10, 1.0), np.arange(0.5, 360, 4.0)
tas = rng.standard_normal((3, 1, len(lat), len(lon0)))
mk = lambda m: m.Dataset({"tas": (("time", "z", "latitude", "long"), tas)},
                         coords={"time": np.arange(3), "z": np.array([2.0]), "latitude": lat, "long": lon0})
dx, dm = mk(xr), mk(minixr)
assert A._is_xarray(dx) and not A._is_xarray(dm)
rx, rm = rename_coords_to_lon_and_lat(dx), rename_coords_to_lon_and_lat(dm)      # utils.py:43-57: rename, drop z, squeeze
assert isinstance(rx, xr.Dataset) and tuple(rx["tas"].dims) == tuple(rm["tas"].dims) == ("time", "lat", "lon")
assert "z" not in rx.coords and "z" not in rm.coords
sx, sm = standardize_climate_data(dx), standardize_climate_data(dm)
np.testing.assert_array_equal(sx["lon"].values, sm.coords["lon"].values)
np.testing.assert_array_equal(sx["tas"].values, sm["tas"].values)
mx, mm = convert_lons_mono(sx, "lon"), convert_lons_mono(sm, "lon")                 # utils.py:23-30
np.testing.assert_array_equal(mx["lon"].values, mm.coords["lon"].values)
np.testing.assert_array_equal(mx["tas"].values, mm["tas"].values)
vals, dims, coords, was_xr = A._extract(sx)                                       # aggregations._extract on an xarray object
assert was_xr and dims["tas"] == ("time", "lat", "lon") and isinstance(coords["lon"], minixr.DataArray)
assert A._lon_perms(sx) == {} and A._xforms(sx) == {} and A._edds(sx) == {}
out = A._as_dataset({"tas": np.zeros((3, 2))}, ("time", "hierid"), {"time": np.arange(3), "hierid": np.array(["a", "b"], dtype=object)}, True)
assert isinstance(out, xr.Dataset) and tuple(out["tas"].dims) == ("time", "hierid") and list(out["hierid"].values) == ["a", "b"]
print("xarray stand-in (host): ok")
'''


def test_xarray_branches_of_the_host_code_meet_a_stand_in():
    """The xarray routes of standardize.py and the xarray ends of aggregations.py (`_is_xarray`, `_extract`, `_as_dataset`),
    executed in a child process against tests/stubs/xarray -- a TESTS-ONLY stand-in with exactly the attributes the package
    touches (xarray itself is not installed in this image) -- and compared with the minixr route.  No GPU needed: nothing
    here aggregates (tests/test_gpu_round5.py runs the drop-in on such objects)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "tests", "stubs"), root, os.environ.get("PYTHONPATH", "")]))
    p = subprocess.run([sys.executable, "-c", _XARRAY_HOST_CHILD], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert p.returncode == 0 and "xarray stand-in (host): ok" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
