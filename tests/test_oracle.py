"""Pins the oracle: fixture replay, three independent restatements, known answers, properties.
CPU only.  Follows SURVEY.md section 4.4 / 8c (the reference pins no numbers itself)."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import c_oracle, ref_numpy as O

DIMS_FIX = ("lat", "lon", "time")


def _args(fx, wname, lname):
    return (fx["temp"], DIMS_FIX, fx["lat"], fx["lon"], fx["seg_lat"], fx["seg_lon"], fx[wname],
            fx["areawt"], fx[lname])


@pytest.mark.parametrize("wname", ["popwt", "areawt"])
@pytest.mark.parametrize("lname", ["ISO", "hierid"])
def test_fixture_three_restatements_agree(ref_fixture, wname, lname):
    fx, gold = ref_fixture
    a, da, ua = O.agg_scatter(*_args(fx, wname, lname), group_dim=lname)
    b, db, ub = O.agg_pandas(*_args(fx, wname, lname), group_dim=lname)
    c, dc, uc = O.agg_csr(*_args(fx, wname, lname), group_dim=lname)
    assert da == db == dc == (lname, "time")        # S10, cf. tests/test_climate_toolbox.py:115
    assert list(ua) == list(ub) == list(uc) == list(gold["labels_%s" % lname])
    np.testing.assert_allclose(b, a, rtol=1e-12)
    np.testing.assert_allclose(c, a, rtol=1e-12)
    np.testing.assert_allclose(a, gold["expect_%s_%s" % (wname, lname)], rtol=1e-13)
    assert not np.isnan(a).any()                     # tests/test_climate_toolbox.py:129,135


def test_fixture_provisional_values(ref_fixture):
    """SURVEY 8c provisional t=0 values (restatement-derived, not xarray output)."""
    _, gold = ref_fixture
    np.testing.assert_allclose(gold["expect_popwt_ISO"][:, 0],
                               [58.62485545, 47.52367874, 36.92702391, 58.43411999], rtol=1e-9)
    np.testing.assert_allclose(gold["expect_areawt_ISO"][:, 0],
                               [53.55318525, 47.07807819, 38.83228208, 57.99791332], rtol=1e-9)


def test_c_oracle_matches_numpy_on_fixture(ref_fixture):
    fx, gold = ref_fixture
    X = np.ascontiguousarray(fx["temp"].reshape(-1, fx["temp"].shape[-1]))   # (G, T)
    ilat = O._lookup_dict(fx["lat"], fx["seg_lat"], "lat")
    ilon = O._lookup_dict(fx["lon"], fx["seg_lon"], "lon")
    cell = ilat * len(fx["lon"]) + ilon
    for wname in ("popwt", "areawt"):
        w_eff = O.effective_weights(fx[wname], fx["areawt"])
        uniq, code = O.region_codes(fx["ISO"])
        got = c_oracle.segments(X, cell, code, w_eff, len(uniq), layout="GT")   # (T, R)
        np.testing.assert_allclose(got.T, gold["expect_%s_ISO" % wname], rtol=1e-12)
        got2 = O.agg_coded(X.T, cell, code, w_eff, len(uniq))
        np.testing.assert_allclose(got2.T, gold["expect_%s_ISO" % wname], rtol=1e-12)


def test_known_answers(golden_dir):
    k = np.load(os.path.join(golden_dir, "kat_small.npz"), allow_pickle=False)
    lab = np.array([None if n else str(s) for s, n in zip(k["lab"], k["lab_null"])], dtype=object)
    for wname in ("areawt", "popwt"):
        for fn in (O.agg_scatter, O.agg_pandas, O.agg_csr):
            got, dims, labs = fn(k["X"], ("time", "lat", "lon"), k["lat"], k["lon"], k["seg_lat"],
                                 k["seg_lon"], k[wname], k["areawt"], lab, group_dim="lab")
            assert dims == ("time", "lab")
            assert list(labs) == list(k["labels"])             # S3 sorted, null label dropped
            np.testing.assert_allclose(got, k["expect_%s" % wname], rtol=1e-14)


def test_missing_label_raises_keyerror():
    lat, lon = np.array([0.0, 1.0]), np.array([0.0, 1.0])
    X = np.zeros((1, 2, 2))
    for fn in (O.agg_scatter, O.agg_pandas, O.agg_csr):
        with pytest.raises(KeyError):                          # S1
            fn(X, ("time", "lat", "lon"), lat, lon, [0.5], [0.0], [1.0], [1.0], ["a"])


def test_zero_denominator_and_all_nan_region():
    lat, lon = np.array([0.0]), np.array([0.0, 1.0, 2.0])
    X = np.array([[[np.nan, 2.0, 3.0]]])
    got, _, labs = O.agg_scatter(X, ("time", "lat", "lon"), lat, lon, [0.0, 0.0, 0.0, 0.0],
                                 [0.0, 1.0, 1.0, 2.0], [1.0, 1.0, -1.0, np.nan],
                                 [2.0, 1.0, -1.0, np.nan], ["nan_cell", "zero_den", "zero_den", "nan_w"])
    assert list(labs) == ["nan_cell", "nan_w", "zero_den"]
    assert got[0, 0] == 0.0                                    # S6: all-NaN region -> 0/den = 0
    assert np.isnan(got[0, 1])                                 # NaN weight leaves both sums: 0/0
    assert np.isnan(got[0, 2])                                 # 1 + (-1) = 0 denominator, (2-2)/0 (S7)


def test_dim_order_S10():
    lat, lon = np.array([0.0, 1.0]), np.array([5.0, 6.0, 7.0])
    rng = np.random.default_rng(0)
    base = rng.random((4, 2, 3))
    seg = dict(seg_lat=[0.0, 1.0, 1.0], seg_lon=[5.0, 7.0, 6.0], w=[1.0, 2.0, 3.0], lab=[1, 1, 2])
    ref, d0, _ = O.agg_scatter(base, ("time", "lat", "lon"), lat, lon, seg["seg_lat"], seg["seg_lon"],
                               seg["w"], seg["w"], seg["lab"], group_dim="g")
    assert d0 == ("time", "g")
    for perm, dims in (((1, 2, 0), ("lat", "lon", "time")), ((2, 1, 0), ("lon", "lat", "time")),
                       ((1, 0, 2), ("lat", "time", "lon"))):
        got, d, _ = O.agg_scatter(np.transpose(base, perm), dims, lat, lon, seg["seg_lat"],
                                  seg["seg_lon"], seg["w"], seg["w"], seg["lab"], group_dim="g")
        assert d == tuple("g" if x == dims[min(dims.index("lat"), dims.index("lon"))] else x
                          for x in dims if x != dims[max(dims.index("lat"), dims.index("lon"))])
        np.testing.assert_allclose(np.moveaxis(got, d.index("g"), -1), ref, rtol=1e-14)


@st.composite
def _case(draw):
    G = draw(st.integers(2, 40))
    R = draw(st.integers(1, 6))
    n = draw(st.integers(1, 80))
    T = draw(st.integers(1, 5))
    seed = draw(st.integers(0, 2**31 - 1))
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((T, G)) * 10, rng.integers(0, G, n), rng.integers(0, R, n),
            rng.uniform(0.1, 2.0, n), R, rng)


@settings(max_examples=60, deadline=None)
@given(_case())
def test_properties(case):
    X, cell, code, w, R, rng = case
    ref = O.agg_coded(X, cell, code, w, R)
    p = rng.permutation(len(cell))                                     # row order is irrelevant (S12)
    np.testing.assert_allclose(O.agg_coded(X, cell[p], code[p], w[p], R), ref, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(O.agg_coded(3.0 * X + 0.0, cell, code, w, R), 3.0 * ref, rtol=1e-10,
                               atol=1e-12)                             # linear in X
    const = O.agg_coded(np.full_like(X, 7.25), cell, code, w, R)
    present = np.bincount(code, minlength=R) > 0
    np.testing.assert_allclose(const[:, present], 7.25, rtol=1e-12)     # constant field -> constant
    scale = np.where(code == code[0], 5.0, 1.0)                        # rescaling one region's weights
    np.testing.assert_allclose(O.agg_coded(X, cell, code, w * scale, R), ref, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(c_oracle.segments(X, cell, code, w, R), ref, rtol=1e-10, atol=1e-12)


def test_dense_hash_and_dense_oracle_agree():
    G, R, seed = 37, 11, 5
    W = O.dense_weights_oracle(G, R, seed)
    L = c_oracle.lib()
    for g, r in ((0, 0), (3, 7), (36, 10)):
        assert W[g, r] == L.wagg_oracle_hash_u01(g * R + r, seed)
    assert (W >= 0).all() and (W < 1).all()
    X = np.random.default_rng(1).standard_normal((5, G)).astype(np.float32)
    X[2, 3] = np.nan
    ref = O.agg_dense(X, W)
    got = c_oracle.dense_synth(X, 0, G, R, 0, R, seed)
    np.testing.assert_allclose(got, ref, rtol=1e-12)
    win = c_oracle.dense_synth(X, 0, G, R, 4, 5, seed)                 # column window
    np.testing.assert_allclose(win, ref[:, 4:9], rtol=1e-12)


def test_oracle_reproduces_the_numbers_the_reference_tests_hold(golden_dir):
    """tests/golden/reference_held.npz = the numeric expectations of the reference's own tests for
    the helpers either side of the path (tests/test_climate_toolbox.py:177-290)."""
    h = np.load(os.path.join(golden_dir, "reference_held.npz"), allow_pickle=False)
    np.testing.assert_array_equal(O.convert_lons_mono_labels(h["mono_lon"]), h["mono_expect"])          # :180-187
    vals, labs = O.convert_lons_split(np.arange(2.0), ("lon",), h["split_lon"])
    np.testing.assert_array_equal(labs, h["split_expect"])                                               # :190-197
    np.testing.assert_array_equal(vals, np.arange(2.0))
    edd = O.snyder_edd_values(h["edd_tmin"], h["edd_tmax"], float(h["edd_threshold"]))
    assert edd.sum() == float(h["edd_sum"]) == 0.0                                                       # :262
    gdd = O.snyder_gdd_values(h["edd_tmin"], h["edd_tmax"], float(h["gdd_threshold_low"]),
                              float(h["gdd_threshold_high"]))
    assert gdd.sum() == pytest.approx(float(h["gdd_sum_approx"]), float(h["gdd_sum_rel"]))               # :290
    # fp32 fields give the same answers (both cells lie in the "tmax < e" / "tmin >= e" branches)
    assert O.snyder_edd_values(h["edd_tmin"].astype(np.float32), h["edd_tmax"].astype(np.float32),
                               float(h["edd_threshold"])).sum() == 0.0


def test_c_oracle_csr_generator_is_the_numpy_hash_table():
    """oracle/wagg_oracle.c::wagg_oracle_synth_csr (the caller-side arrays of the full-size c5 parity test, vectorised
    integer form of the hash test) against the NumPy restatement of the same hashes, both structures, and across the
    row where the 64-bit counter g R + r crosses 2^32."""
    G, R = 5000, 300
    rp, col, val = c_oracle.synth_csr(G, R, 7, 0.02)
    W = O.dense_weights_oracle(G, R, 7, 0.02)
    gi, ri = np.nonzero(W)
    np.testing.assert_array_equal(np.repeat(np.arange(G), np.diff(rp)), gi)
    np.testing.assert_array_equal(col, ri)
    np.testing.assert_array_equal(val, W[gi, ri].astype(np.float64))
    Gs, Rs = 64 * 30 + 7, 700
    Wb = O.blocklocal_weights_oracle(Gs, Rs, 5)
    rp, col, val = c_oracle.synth_csr(Gs, Rs, 5, 0.952, blocklocal=True)
    gi, ri = np.nonzero(Wb)
    np.testing.assert_array_equal(np.repeat(np.arange(Gs), np.diff(rp)), gi)
    np.testing.assert_array_equal(col, ri)
    np.testing.assert_array_equal(val, Wb[gi, ri].astype(np.float64))
    G, R, g0 = 720 * 1440, 24378, 176170                          # g R = 2^32 inside row 176,181
    rp, col, val = c_oracle.synth_csr(G, R, 2, 0.01, g0=g0, g1=g0 + 30)
    for g in (176180, 176181, 176182):
        idx = np.uint64(g) * np.uint64(R) + np.arange(R, dtype=np.uint64)
        keep = O.hash_u01(idx, np.uint32(2) ^ np.uint32(0x9e3779b9)) < np.float32(0.01)
        a, b = rp[g - g0], rp[g - g0 + 1]
        np.testing.assert_array_equal(col[a:b], np.flatnonzero(keep))
        np.testing.assert_array_equal(val[a:b], O.hash_u01(idx, 2)[keep].astype(np.float64))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_threaded_c_oracle_equals_the_single_threaded_one(dtype):
    """bench.py's best-effort CPU leg (timesteps dealt to OpenMP threads) gives the faithful port's bits."""
    rng = np.random.default_rng(4)
    T, G, R, nseg = 37, 900, 40, 5000
    X = rng.standard_normal((T, G)).astype(dtype)
    X[3, 10] = np.nan
    ci = rng.integers(0, G, nseg)
    rc = rng.integers(-1, R, nseg)
    w = rng.uniform(0, 1, nseg)
    w[::17] = np.nan
    a = c_oracle.segments(X, ci, rc, w, R)
    b = c_oracle.segments(X, ci, rc, w, R, threaded=True)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(c_oracle.segments(X.T.copy(), ci, rc, w, R, layout="GT", threaded=True), a)
