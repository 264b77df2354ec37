"""GPU parity tests (run with -m gpu on an MI355X).  Every comparison goes through the C-ABI
(libwagg.so); the oracle is only the checker.  Tolerances: fp64 1e-6 relative, fp32 1e-4
relative (BASELINE.json north_star), written next to each assertion."""
import os
import sys

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu

RTOL64, RTOL32 = 1e-6, 1e-4


def _rel_ok(got, ref, rtol, scale=None):
    """|got - ref| <= rtol * max(|ref|, scale) per region-timestep; NaN/inf must match exactly."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape
    fin = np.isfinite(ref)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)])
    s = np.abs(ref[fin]) if scale is None else np.maximum(np.abs(ref[fin]), scale)
    err = np.abs(got[fin] - ref[fin])
    bad = err > rtol * s + 1e-300
    assert not bad.any(), "max rel err %.3e at %d of %d" % ((err / np.maximum(s, 1e-300)).max(), bad.sum(), bad.size)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


def _mk_ds(fx):
    from climate_toolbox_amd import minixr
    return minixr.Dataset({"temperature": (["lat", "lon", "time"], fx["temp"])},
                          coords={"lon": fx["lon"], "lat": fx["lat"], "time": np.arange(10)})


def _mk_weights(fx):
    df = pd.DataFrame({"lat": fx["seg_lat"], "lon": fx["seg_lon"], "areawt": fx["areawt"],
                       "popwt": fx["popwt"], "hierid": fx["hierid"], "ISO": fx["ISO"]})
    df.index.names = ["reshape_index"]
    return df


# ---------------------------------------------------------------------------------------------
# the reference's own tests, replayed through the drop-in (tests/test_climate_toolbox.py:109-135)
# ---------------------------------------------------------------------------------------------
def test_reference_test_reindex_spatial_weights(ref_fixture, torch_cuda):
    from climate_toolbox_amd import _reindex_spatial_data_to_regions
    fx, _ = ref_fixture
    clim_data, weights = _mk_ds(fx), _mk_weights(fx)
    assert not clim_data.temperature.isnull().any()
    ds = _reindex_spatial_data_to_regions(clim_data, weights)
    assert ds.temperature.shape == (len(ds["lon"]), len(ds["time"]))
    assert "reshape_index" in ds.dims
    # the materialised gather equals numpy pointwise indexing (bit exact: it is a copy)
    ilat = np.searchsorted(fx["lat"], fx["seg_lat"])
    ilon = np.searchsorted(fx["lon"], fx["seg_lon"])
    np.testing.assert_array_equal(ds.temperature.values, fx["temp"][ilat, ilon, :])


def test_reference_test_weighting(ref_fixture, torch_cuda):
    from climate_toolbox_amd import _aggregate_reindexed_data_to_regions, _reindex_spatial_data_to_regions
    fx, gold = ref_fixture
    clim_data, weights = _mk_ds(fx), _mk_weights(fx)
    assert np.isnan(weights["popwt"].values).any()
    ds = _reindex_spatial_data_to_regions(clim_data, weights)
    assert not ds.temperature.isnull().any()
    for wname in ("popwt", "areawt"):
        wtd = _aggregate_reindexed_data_to_regions(ds, "temperature", wname, "ISO", weights)
        assert not wtd.temperature.isnull().any()
        assert wtd.temperature.dims == ("ISO", "time")                       # S10
        np.testing.assert_array_equal(wtd["ISO"].values, gold["labels_ISO"])  # S3
        _rel_ok(wtd.temperature.values, gold["expect_%s_ISO" % wname], RTOL64)  # fp64 data -> 1e-6


@pytest.mark.parametrize("wname", ["popwt", "areawt"])
@pytest.mark.parametrize("lname", ["ISO", "hierid"])
@pytest.mark.parametrize("dtype,rtol", [(np.float64, RTOL64), (np.float32, RTOL32)])
def test_weighted_aggregate_grid_to_regions_fixture(ref_fixture, torch_cuda, wname, lname, dtype, rtol):
    from climate_toolbox_amd import minixr, weighted_aggregate_grid_to_regions
    fx, gold = ref_fixture
    ds = minixr.Dataset({"temperature": (["lat", "lon", "time"], fx["temp"].astype(dtype))},
                        coords={"lon": fx["lon"], "lat": fx["lat"], "time": np.arange(10)})
    out = weighted_aggregate_grid_to_regions(ds, "temperature", wname, lname, _mk_weights(fx))
    assert out.temperature.dims == (lname, "time")
    _rel_ok(out.temperature.values, gold["expect_%s_%s" % (wname, lname)], rtol)
    # (time, lat, lon) files: same numbers, (time, region) order
    ds2 = minixr.Dataset({"temperature": (["time", "lat", "lon"],
                                          np.ascontiguousarray(np.moveaxis(fx["temp"], -1, 0)).astype(dtype))},
                         coords={"lon": fx["lon"], "lat": fx["lat"], "time": np.arange(10)})
    out2 = weighted_aggregate_grid_to_regions(ds2, "temperature", wname, lname, _mk_weights(fx))
    assert out2.temperature.dims == ("time", lname)
    _rel_ok(out2.temperature.values.T, gold["expect_%s_%s" % (wname, lname)], rtol)


def test_known_answers_through_dropin(golden_dir, torch_cuda):
    from climate_toolbox_amd import minixr, weighted_aggregate_grid_to_regions
    k = np.load(os.path.join(golden_dir, "kat_small.npz"), allow_pickle=False)
    lab = np.array([None if n else str(s) for s, n in zip(k["lab"], k["lab_null"])], dtype=object)
    df = pd.DataFrame({"lat": k["seg_lat"], "lon": k["seg_lon"], "areawt": k["areawt"],
                       "popwt": k["popwt"], "lab": lab})
    ds = minixr.Dataset({"v": (("time", "lat", "lon"), k["X"])}, coords={"lat": k["lat"], "lon": k["lon"]})
    for wname in ("areawt", "popwt"):
        out = weighted_aggregate_grid_to_regions(ds, "v", wname, "lab", df)
        assert list(out["lab"].values) == list(k["labels"])
        _rel_ok(out.v.values, k["expect_%s" % wname], 1e-14)       # tiny sums: exact to rounding
    bad = df.copy()
    bad.loc[0, "lon"] = 105.0
    with pytest.raises(KeyError):                                   # S1
        weighted_aggregate_grid_to_regions(ds, "v", "areawt", "lab", bad)


# ---------------------------------------------------------------------------------------------
# coded-table cases straight on the C-ABI objects
# ---------------------------------------------------------------------------------------------
def _random_case(rng, T, G, R, nseg, giant=0, row_len=0):
    cell = rng.integers(0, G, nseg)
    code = rng.integers(0, R, nseg)
    if giant:                      # a few regions covering thousands of cells (multi-chunk path)
        for r in range(giant):
            n = int(rng.integers(300, 3000))
            start = int(rng.integers(0, G - n))
            cell = np.concatenate([cell, np.arange(start, start + n)])
            code = np.concatenate([code, np.full(n, r)])
    w = rng.uniform(0.05, 3.0, len(cell))
    X = 250.0 + 50.0 * rng.standard_normal((T, G))
    return X, cell.astype(np.int32), code.astype(np.int32), w


@pytest.mark.parametrize("T,G,R,nseg,giant", [(1, 50, 3, 40, 0), (10, 1000, 37, 3000, 0),
                                              (64, 5000, 200, 20000, 2), (65, 5000, 200, 20000, 1),
                                              (130, 20000, 50, 5000, 3), (33, 300, 1, 299, 0)])
@pytest.mark.parametrize("dtype,rtol", [(np.float64, RTOL64), (np.float32, RTOL32)])
def test_sparse_random_cases(torch_cuda, T, G, R, nseg, giant, dtype, rtol):
    from climate_toolbox_amd.engine import SparsePlan
    from oracle import c_oracle, ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(T * 1000 + G + nseg + giant)
    X, cell, code, w = _random_case(rng, T, G, R, nseg, giant)
    X = X.astype(dtype)
    X[rng.integers(0, T), rng.integers(0, G)] = np.nan            # S6
    ref = O.agg_coded(X, cell, code, w, R)
    np.testing.assert_allclose(c_oracle.segments(X, cell, code, w, R), ref, rtol=1e-11)   # oracle vs oracle
    plan = SparsePlan(cell, code, w, G, R, row_len=100)
    assert plan.info["n_giant"] >= giant - 0 if giant else True
    np.testing.assert_allclose(plan.den, np.bincount(code, weights=w, minlength=R), rtol=1e-13)
    Xd = torch.from_numpy(X).cuda()
    got = plan.apply(Xd).cpu().numpy()
    _rel_ok(got, ref, rtol, scale=1.0)
    got_rt = plan.apply(Xd, out_layout="RT").cpu().numpy()
    np.testing.assert_array_equal(got_rt.T, got)                  # same kernel arithmetic, bitwise
    Xgt = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    got_gt = plan.apply(Xgt, layout="GT").cpu().numpy()
    # (time, gridcell) data may take the whole-line chunking (partial sums per (chunk, region), fused multiply-adds);
    # (gridcell, time) data the region-shaped chunks: another summation order (S12), not another result
    _rel_ok(got_gt, got, 1e-13 if dtype == np.float64 else 2e-6, scale=1.0)
    again = plan.apply(Xd).cpu().numpy()
    np.testing.assert_array_equal(again, got)                     # no atomics: reproducible
    host = plan.apply_host(X)                                     # blocking host-buffer ABI form
    np.testing.assert_array_equal(host, got)


def test_sparse_edge_semantics(torch_cuda):
    """S3-S7 on the device: null labels, NaN/zero/negative weights after the backup fill, NaN and
    inf data, zero denominators, regions without any kept row, duplicate rows."""
    from climate_toolbox_amd.engine import SparsePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    G, R = 12, 7
    X = np.arange(3 * G, dtype=np.float64).reshape(3, G) + 1.0
    X[1, 2] = np.nan
    X[2, 5] = np.inf
    cell = np.array([0, 1, 2, 2, 3, 4, 5, 5, 6, 7, 8, 9, 9], dtype=np.int32)
    code = np.array([0, 0, 1, 1, -1, 2, 3, 3, 4, 4, 5, 6, 6], dtype=np.int32)
    w = np.array([1.0, 2.0, 0.5, 0.5, 9.0, np.nan, 0.0, 2.0, 1.0, -1.0, -2.0, 0.0, 0.0])
    ref = O.agg_coded(X, cell, code, w, R)
    for dtype, rtol in ((np.float64, 1e-14), (np.float32, 1e-6)):
        plan = SparsePlan(cell, code, w, G, R)
        got = plan.apply(torch.from_numpy(X.astype(dtype)).cuda()).cpu().numpy()
        _rel_ok(got, ref, rtol)
    assert np.isnan(ref[:, 2]).all()          # only a NaN-weight row: 0/0 (region without kept row)
    assert ref[1, 1] == 0.0                   # all-NaN data region -> 0/den = 0   (S6)
    assert np.isposinf(ref[2, 3])             # inf*0 is a NaN product -> skipped; inf*2/2 = inf
    assert np.isneginf(ref[:, 4]).all()       # 1 + (-1) = 0 denominator, negative numerator (S7)
    assert (ref[:, 5] > 0).all()              # negative weight alone: (-2x)/(-2) = x
    assert np.isnan(ref[:, 6]).all()          # zero weights: 0/0


def test_empty_and_degenerate(torch_cuda):
    from climate_toolbox_amd.engine import SparsePlan
    torch = torch_cuda
    plan = SparsePlan(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), 10, 3)
    out = plan.apply(torch.ones((4, 10), dtype=torch.float32, device="cuda")).cpu().numpy()
    assert out.shape == (4, 3) and np.isnan(out).all()             # every region empty: 0/0
    out0 = plan.apply(torch.ones((0, 10), dtype=torch.float64, device="cuda"))
    assert tuple(out0.shape) == (0, 3)
    with pytest.raises(Exception):
        SparsePlan(np.array([11], np.int32), np.array([0], np.int32), np.ones(1), 10, 3)   # cell out of range
    with pytest.raises(Exception):
        SparsePlan(np.array([1], np.int32), np.array([3], np.int32), np.ones(1), 10, 3)    # code out of range


def test_c1_full(torch_cuda):
    """BASELINE.json configs[0]: 1-year daily, 2-degree grid, 100 regions, fp64, area- and pop-weighted."""
    from climate_toolbox_amd import minixr, weighted_aggregate_grid_to_regions, synth
    from oracle import ref_numpy as O
    lat, lon, tas, df = synth.c1_workload(T=365)
    ds = minixr.Dataset({"tas": (("time", "lat", "lon"), tas)}, coords={"lat": lat, "lon": lon})
    for wname in ("areawt", "popwt"):
        ref, dims, labs = O.agg_scatter(tas, ("time", "lat", "lon"), lat, lon, df["lat"].values,
                                        df["lon"].values, df[wname].values, df["areawt"].values,
                                        df["hierid"].values, group_dim="hierid")
        out = weighted_aggregate_grid_to_regions(ds, "tas", wname, "hierid", df)
        assert out.tas.dims == dims and list(out["hierid"].values) == list(labs)
        _rel_ok(out.tas.values, ref, RTOL64)


# ---------------------------------------------------------------------------------------------
# dense path
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,G,R", [(5, 64, 7), (365, 1000, 130), (368, 4099, 257), (400, 777, 129), (3, 20, 600), (31, 500, 300), (100, 333, 20), (700, 256, 520), (17, 96, 257)])
def test_dense_small_vs_oracle(torch_cuda, T, G, R):
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(T + G + R)
    W = rng.uniform(0, 1, (G, R)).astype(np.float32)
    X = (250 + 50 * rng.standard_normal((T, G))).astype(np.float32)
    X[rng.integers(0, T), rng.integers(0, G)] = np.nan
    ref = O.agg_dense(X, W)
    plan = DensePlan.from_host(W)
    np.testing.assert_allclose(plan.den, W.astype(np.float64).sum(0), rtol=1e-12)
    for ks in (0, 8, 16):
        got = plan.apply(torch.from_numpy(X).cuda(), ksplit=ks).cpu().numpy()
        _rel_ok(got, ref, RTOL32)
    # A = I check with an asymmetric B: catches a transposed C/D fragment map
    if T <= G:
        Xi = np.zeros((T, G), dtype=np.float32)
        Xi[np.arange(T), np.arange(T)] = 1.0
        goti = plan.apply(torch.from_numpy(Xi).cuda()).cpu().numpy()
        _rel_ok(goti, W[:T].astype(np.float64) / W.astype(np.float64).sum(0)[None, :], 1e-6)


def test_dense_synth_matches_oracle_hash(torch_cuda):
    from climate_toolbox_amd.engine import DensePlan, synth_field
    from oracle import c_oracle, ref_numpy as O
    torch = torch_cuda
    G, R, seed = 3000, 300, 2
    plan = DensePlan.synth(G, R, seed)
    W = O.dense_weights_oracle(G, R, seed)
    np.testing.assert_allclose(plan.den, W.astype(np.float64).sum(0), rtol=1e-12)
    X = synth_field(40, G, 7, 280.0, 60.0)
    Xh = X.cpu().numpy()
    idx = (np.arange(40)[:, None] * G + np.arange(G)[None, :]).astype(np.uint64)
    np.testing.assert_array_equal(Xh, (np.float32(280.0) + np.float32(60.0) * (O.hash_u01(idx, 7) - np.float32(0.5))))
    got = plan.apply(X).cpu().numpy()
    _rel_ok(got, c_oracle.dense_synth(Xh, 0, G, R, 0, R, seed), RTOL32)
    # c5's uniform-random structure: 5 % of the entries at random positions (still the full dense form)
    sp = DensePlan.synth(G, R, seed, fill=0.05)
    Ws = O.dense_weights_oracle(G, R, seed, fill=0.05)
    assert sp.info["tiled"] == 0 and 0.04 < (Ws != 0).mean() < 0.06
    np.testing.assert_allclose(sp.den, Ws.astype(np.float64).sum(0), rtol=1e-12)
    _rel_ok(sp.apply(X).cpu().numpy(), O.agg_dense(Xh, Ws), RTOL32)


def test_dense_from_segments_equals_sparse(torch_cuda):
    from climate_toolbox_amd.engine import DensePlan, SparsePlan
    torch = torch_cuda
    rng = np.random.default_rng(5)
    X, cell, code, w = _random_case(rng, 40, 2000, 60, 8000)
    Xd = torch.from_numpy(X.astype(np.float32)).cuda()
    a = SparsePlan(cell, code, w, 2000, 60).apply(Xd).cpu().numpy()
    b = DensePlan.from_segments(cell, code, w, 2000, 60).apply(Xd).cpu().numpy()
    _rel_ok(b, a, 2e-4, scale=1.0)


# ---------------------------------------------------------------------------------------------
# BASELINE.json full sizes (configs[1], [2]): direct comparison where the oracle finishes in
# seconds, size-independent properties where it cannot (dense 1,036,800 x 24,378)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2_real():
    from climate_toolbox_amd import synth
    lat, lon, df = synth.realistic_segments()           # 720 x 1440, R = 24,378, ~4e5 segments
    return lat, lon, df


@pytest.mark.parametrize("wname,dtype,rtol", [("areawt", np.float32, RTOL32), ("popwt", np.float64, RTOL64),
                                              ("popwt", np.float32, RTOL32)])
def test_c2_c3_full_size_vs_oracle(torch_cuda, c2_real, wname, dtype, rtol):
    """configs[1] (area-weighted fp32) and configs[2] (pop-weighted fp64 tolerance check, plus the
    fp32 run at 1e-4): every region-timestep against the fp64 oracle."""
    from climate_toolbox_amd import engine, synth
    from oracle import ref_numpy as O
    torch = torch_cuda
    lat, lon, df = c2_real
    T, G = 365, len(lat) * len(lon)
    cell, codes, w_eff, uniq = synth.code_segments(df, lat, lon, wname, "hierid")
    assert len(uniq) == 24378
    if wname == "popwt":
        assert (~(df["popwt"].values > 0)).mean() > 0.15       # S4 really exercised (NaN / 0 rows)
    X = engine.synth_field(T, G, seed=11, base=280.0, amp=60.0, dtype="float64" if dtype == np.float64 else "float32")
    Xh = X.cpu().numpy()
    plan = engine.SparsePlan(cell, codes, w_eff, G, len(uniq), row_len=len(lon))
    got = plan.apply(X).cpu().numpy()
    ref = O.agg_coded(Xh, cell, codes, w_eff, len(uniq))
    _rel_ok(got, ref, rtol)
    # the same field in (gridcell, time) order -- the reference's (lat, lon, time) fixture layout -- at full size
    got_gt = plan.apply(X.t().contiguous(), layout="GT").cpu().numpy()
    _rel_ok(got_gt, ref, rtol)
    _rel_ok(got_gt, got, 2e-6 if dtype == np.float32 else 1e-13)
    del got_gt
    # size-independent properties at full size
    const = plan.apply(torch.full((3, G), 7.25, dtype=X.dtype, device="cuda")).cpu().numpy()
    np.testing.assert_allclose(const, 7.25, rtol=1e-6 if dtype == np.float32 else 1e-12)   # constant field
    got2 = plan.apply(X * 2.0).cpu().numpy()
    np.testing.assert_array_equal(got2, got * dtype(2.0))                                  # exact linearity in 2x
    perm = np.random.default_rng(0).permutation(len(cell))
    plan_p = engine.SparsePlan(cell[perm], codes[perm], w_eff[perm], G, len(uniq), row_len=len(lon))
    np.testing.assert_allclose(plan_p.apply(X).cpu().numpy(), got, rtol=1e-5 if dtype == np.float32 else 1e-12)
    # fused tas_poly at full size (powers 1..3 in one pass): power 1 with offset 0 is the plain aggregation (from the
    # fused kernel's chunking and summation order: equal to rounding, S12), power 2 against the oracle on the squared field
    poly = plan.apply_poly(X, 0.0, 3)
    _rel_ok(poly[0].cpu().numpy(), got, 2e-6 if dtype == np.float32 else 1e-13)
    _rel_ok(poly[0].cpu().numpy(), ref, rtol)
    _rel_ok(poly[1].cpu().numpy(), O.agg_coded(Xh * Xh, cell, codes, w_eff, len(uniq)), rtol)
    del poly


def _tile_windows(R, width, seed):
    """`width` consecutive regions at a random offset inside every 256-region column tile of [0, R)."""
    rng = np.random.default_rng(seed)
    cols = []
    for t0 in range(0, R, 256):
        n = min(256, R - t0)
        off = int(rng.integers(0, max(1, n - width + 1)))
        cols.append(t0 + off + np.arange(min(width, n)))
    return np.concatenate(cols).astype(np.int64)


def test_c2_dense_full_size_properties(torch_cuda):
    """configs[1] in its dense north-star form (101 GB W generated on the device).  The fp64 oracle
    cannot do 1.8e13 MACs, so: a column window against the C oracle on all 1,036,800 cells, the
    denominators against the hash, constant-field and scaling properties over all 24,378 regions."""
    from climate_toolbox_amd import engine
    from oracle import c_oracle, ref_numpy as O
    torch = torch_cuda
    G, R, seed = 720 * 1440, 24378, 2
    try:
        plan = engine.DensePlan.synth(G, R, seed)
    except Exception as e:                      # a box without 110 GB free HBM cannot hold the operand
        pytest.skip("dense W does not fit: %s" % e)
    T = 24
    X = engine.synth_field(T, G, seed=5, base=280.0, amp=60.0)
    got = plan.apply(X).cpu().numpy()
    Xh = X.cpu().numpy()
    # one 16-column window in EVERY one of the 96 column tiles (random offset inside the tile, fixed seed; the last
    # tile holds 58 regions), all 1,036,800 cells each: a permutation inside any tile would show
    cols = _tile_windows(R, 16, seed=20)
    assert len(cols) == 96 * 16 and len(set(cols // 256)) == 96
    _rel_ok(got[:, cols], c_oracle.dense_synth_cols(Xh, G, R, cols, seed), RTOL32)
    idx = (np.arange(G, dtype=np.uint64) * np.uint64(R) + np.uint64(777))
    np.testing.assert_allclose(plan.den[777], O.hash_u01(idx, seed).astype(np.float64).sum(), rtol=1e-12)
    const = plan.apply(torch.full((2, G), 3.5, dtype=torch.float32, device="cuda")).cpu().numpy()
    np.testing.assert_allclose(const, 3.5, rtol=2e-5)                   # constant field -> constant
    np.testing.assert_array_equal(plan.apply(X * 2.0).cpu().numpy(), got * np.float32(2.0))
    # all rows of a 365-row block agree with the same rows computed in a short block
    X365 = engine.synth_field(365, G, seed=5, base=280.0, amp=60.0)
    got365 = plan.apply(X365).cpu().numpy()
    np.testing.assert_array_equal(got365[:T], got)
    # ... and one row of EVERY one of the 23 sixteen-row MFMA blocks of that launch (a different row inside each), the same
    # 96 column windows, all 1,036,800 cells, against the C oracle (rows 0-23 alone touch two blocks)
    rows = np.array([16 * m + (7 * m) % 16 for m in range(23)])
    assert rows.max() < 365 and len(set(rows // 16)) == 23
    _rel_ok(got365[rows][:, cols], c_oracle.dense_synth_cols(X365[torch.from_numpy(rows).cuda()].cpu().numpy(), G, R, cols, seed), RTOL32)
    del X365, got365
    # configs[3] in the dense form: one rank's 1,369-row shard = four row blocks, whose first pass reads X in place
    # (no packed copy); same bits for the shared rows, the last rows against the C oracle
    X1369 = engine.synth_field(1369, G, seed=5, base=280.0, amp=60.0)
    got1369 = plan.apply(X1369)
    np.testing.assert_array_equal(got1369[:T].cpu().numpy(), got)
    tail = X1369[1360:1369].cpu().numpy()
    for r0 in (0, R - 16):
        _rel_ok(got1369[1360:1369, r0:r0 + 16].cpu().numpy(), c_oracle.dense_synth(tail, 0, G, R, r0, 16, seed), RTOL32)
    assert not plan.saw_inf()
    plan.close()


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[4] shape at test scale: ensemble x time rows, CSR-like weights with a
# fixed fraction of non-zeros, both structures SURVEY 8d names (uniform-random columns = every
# region is a "giant" multi-chunk group; block-local = each 64-cell run touches few regions)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("structure", ["uniform", "block_local"])
@pytest.mark.parametrize("dtype,rtol", [(np.float32, RTOL32), (np.float64, RTOL64)])
def test_c5_like_sparse_structures(torch_cuda, structure, dtype, rtol):
    from climate_toolbox_amd.engine import DensePlan, SparsePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(7)
    members, days, G, R = 5, 37, 64 * 96, 300                # rows = ensemble x time = 185
    T = members * days
    nnz = int(0.01 * G * R)
    if structure == "uniform":
        flat = rng.choice(G * R, size=nnz, replace=False)
        cell, code = (flat // R).astype(np.int32), (flat % R).astype(np.int32)
    else:
        runs = rng.integers(0, G // 64, nnz)
        cell = (runs * 64 + rng.integers(0, 64, nnz)).astype(np.int32)
        code = ((runs * 3 + rng.integers(0, 3, nnz)) % R).astype(np.int32)     # <= 3 regions per run
    w = rng.uniform(0.1, 1.0, nnz)
    X = (280 + 15 * rng.standard_normal((T, G))).astype(dtype)
    ref = O.agg_coded(X, cell, code, w, R)
    plan = SparsePlan(cell, code, w, G, R, row_len=96)
    if structure == "uniform":
        assert plan.info["n_giant"] > 0                       # multi-chunk path exercised
    got = plan.apply(torch.from_numpy(X).cuda()).cpu().numpy()
    _rel_ok(got, ref, rtol, scale=1.0)
    # ensemble x time is just more rows: aggregating members separately gives the same numbers
    one = plan.apply(torch.from_numpy(np.ascontiguousarray(X[days:2 * days])).cuda()).cpu().numpy()
    np.testing.assert_array_equal(one, got[days:2 * days])
    if dtype == np.float32:                                   # the dense-tile form of the same weights
        dense = DensePlan.from_segments(cell, code, w, G, R)
        _rel_ok(dense.apply(torch.from_numpy(X).cuda()).cpu().numpy(), ref, RTOL32, scale=1.0)


def test_dropin_switches_to_dense_form_for_scattered_weights(torch_cuda):
    """Through the reference-named API: weights whose regions are scattered over the whole grid
    are contracted by the dense MFMA kernel, compact ones by the gather kernel -- same numbers."""
    from climate_toolbox_amd import aggregations as A, minixr
    from climate_toolbox_amd.engine import DensePlan, SparsePlan
    from oracle import ref_numpy as O
    rng = np.random.default_rng(3)
    nlat, nlon, R, T = 48, 96, 40, 30
    lat, lon = np.arange(nlat) * 1.0, np.arange(nlon) * 1.0
    n = int(0.3 * nlat * nlon * R)                              # dense-ish random table
    flat = rng.choice(nlat * nlon * R, size=n, replace=False)
    cell, lab = flat // R, flat % R
    df = pd.DataFrame({"lat": lat[cell // nlon], "lon": lon[cell % nlon], "areawt": rng.uniform(0.1, 1, n),
                       "hierid": lab})
    tas = (280 + 10 * rng.standard_normal((T, nlat, nlon))).astype(np.float32)
    ds = minixr.Dataset({"tas": (("time", "lat", "lon"), tas)}, coords={"lat": lat, "lon": lon})
    A._PLAN_CACHE.clear()
    out = A.weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
    assert any(isinstance(p, DensePlan) for p in A._PLAN_CACHE.values())
    ref, dims, labs = O.agg_scatter(tas, ("time", "lat", "lon"), lat, lon, df["lat"].values, df["lon"].values,
                                    df["areawt"].values, df["areawt"].values, df["hierid"].values, group_dim="hierid")
    assert out.tas.dims == dims and list(out["hierid"].values) == list(labs)
    _rel_ok(out.tas.values, ref, RTOL32)
    out64 = A.weighted_aggregate_grid_to_regions(
        minixr.Dataset({"tas": (("time", "lat", "lon"), tas.astype(np.float64))}, coords={"lat": lat, "lon": lon}),
        "tas", "areawt", "hierid", df)
    # (round 2: fp64 data takes the fp64 MFMA form of the same table instead of the chunk-walking kernel)
    assert any(isinstance(p, DensePlan) and p.dtype == "float64" for p in A._PLAN_CACHE.values())
    _rel_ok(out64.tas.values, O.agg_scatter(tas.astype(np.float64), ("time", "lat", "lon"), lat, lon, df["lat"].values,
                                            df["lon"].values, df["areawt"].values, df["areawt"].values,
                                            df["hierid"].values)[0], RTOL64)


def test_c4_rank_shard_shapes(torch_cuda, c2_real):
    """BASELINE.json configs[3]: 10,950 daily steps sharded over 8 GPUs = 1,369 / 1,368 rows per
    rank.  One rank's shard at full grid size: sparse fp32 against the oracle on a row sample, and
    shard-by-shard results equal to the unsharded call (rows are independent: the time shard + gather
    of timeshard.py reassembles exactly these blocks)."""
    from climate_toolbox_amd import engine, synth
    from climate_toolbox_amd.timeshard import shard_bounds
    from oracle import ref_numpy as O
    torch = torch_cuda
    lat, lon, df = c2_real
    G = len(lat) * len(lon)
    bounds = shard_bounds(10950, 8)
    assert [e - s for s, e in bounds][:3] == [1369, 1369, 1369] and bounds[-1] == (9582, 10950)
    T = bounds[0][1] - bounds[0][0]                         # 1369 rows: rank 0's shard
    cell, codes, w_eff, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
    plan = engine.SparsePlan(cell, codes, w_eff, G, len(uniq), row_len=len(lon))
    X = engine.synth_field(T, G, seed=21, base=280.0, amp=60.0)
    got = plan.apply(X)
    rows = np.r_[0:3, 700:703, T - 3:T]
    ref = O.agg_coded(X[rows].cpu().numpy(), cell, codes, w_eff, len(uniq))
    _rel_ok(got[rows].cpu().numpy(), ref, RTOL32)
    # two "ranks" splitting this block reproduce it bit for bit
    a, b = shard_bounds(T, 2)
    parts = torch.cat([plan.apply(X[a[0]:a[1]]), plan.apply(X[b[0]:b[1]])], dim=0)
    assert torch.equal(parts, got)


def test_dense_multi_row_block(torch_cuda):
    """Dense form with more than one 368-row block (a c4 rank holds 1,369 rows = 4 blocks)."""
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(11)
    T, G, R = 1369, 1024, 200
    W = rng.uniform(0, 1, (G, R)).astype(np.float32)
    X = (280 + 20 * rng.standard_normal((T, G))).astype(np.float32)
    plan = DensePlan.from_host(W)
    got = plan.apply(torch.from_numpy(X).cuda()).cpu().numpy()
    _rel_ok(got, O.agg_dense(X, W), RTOL32)
    part = plan.apply(torch.from_numpy(X[368:736].copy()).cuda()).cpu().numpy()
    np.testing.assert_array_equal(part, got[368:736])


@pytest.mark.gpu
def test_standardized_raw_0_360_file_matches_reference_order_of_operations(torch_cuda):
    """a raw 0..360 'latitude/longitude' file goes through standardize_climate_data
    (io/io.py:6-24) and the aggregation; the reference permutes the array first, here the
    permutation lives in the plan's cell index -- same numbers."""
    from climate_toolbox_amd import minixr, standardize_climate_data, weighted_aggregate_grid_to_regions
    from oracle import ref_numpy as O
    rng = np.random.default_rng(5)
    lat, lon = np.arange(-89.5, 90, 1.0), np.arange(0.5, 360, 1.0)
    for dims, shape in ((("time", "latitude", "longitude"), (40, 180, 360)),
                        (("latitude", "longitude", "time"), (180, 360, 7))):
        tas = (280 + 10 * rng.standard_normal(shape)).astype(np.float32 if dims[0] == "time" else np.float64)
        ds = minixr.Dataset({"tas": (dims, tas)}, coords={"latitude": lat, "longitude": lon})
        std_dims = tuple({"latitude": "lat", "longitude": "lon"}.get(d, d) for d in dims)
        ref_vals, ref_lon = O.convert_lons_split(tas, std_dims, lon)       # what the reference aggregates
        n = 3000
        df = pd.DataFrame({"lat": rng.choice(lat, n), "lon": rng.choice(ref_lon, n),
                           "areawt": rng.uniform(0.1, 1, n), "popwt": rng.uniform(0, 5, n),
                           "hierid": rng.integers(0, 150, n)})
        df.loc[rng.random(n) < 0.2, "popwt"] = np.nan
        ref, rdims, labs = O.agg_scatter(ref_vals, std_dims, lat, ref_lon, df["lat"].values, df["lon"].values,
                                         df["popwt"].values, df["areawt"].values, df["hierid"].values,
                                         group_dim="hierid")
        out = weighted_aggregate_grid_to_regions(standardize_climate_data(ds), "tas", "popwt", "hierid", df)
        assert out.tas.dims == rdims and list(out["hierid"].values) == list(labs)
        _rel_ok(out.tas.values, ref, RTOL32 if tas.dtype == np.float32 else RTOL64)


@pytest.mark.parametrize("lines", [True, False])
@pytest.mark.parametrize("dtype,layout", [(np.float32, "TG"), (np.float64, "TG"), (np.float32, "GT"), (np.float64, "GT")])
def test_fused_tas_poly_matches_transform_then_aggregate(torch_cuda, dtype, layout, lines):
    """(tas - 273.15) ** p for p = 1..4 fused into the aggregation's loads
    (wagg_apply_poly_*) equals the reference order of operations: transform the grid
    (transformations.py:188), then aggregate (aggregations.py:78-80)."""
    from climate_toolbox_amd import synth
    from climate_toolbox_amd.engine import SparsePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    nlat, nlon, R, T = 96, 192, 300, 77
    lat, lon, df = synth.realistic_segments(nlat, nlon, R=R, seed=4, string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "popwt", "hierid")
    # one region spread over the whole grid -> a multi-chunk ("giant") group rides along
    rng = np.random.default_rng(8)
    extra = rng.choice(nlat * nlon, 900, replace=False).astype(np.int32)
    cell = np.concatenate([cell, extra]); code = np.concatenate([code, np.full(900, len(uniq), np.int32)])
    w = np.concatenate([w, rng.uniform(0.1, 1, 900)])
    Rn, G = len(uniq) + 1, nlat * nlon
    X = (288.15 + 8 * rng.standard_normal((T, G))).astype(dtype)   # ~15 +- 8 degrees C
    X[3, cell[:40]] = np.nan                                      # NaN data: skipped products (S6)
    X[5, cell[100]], X[6, cell[101]], X[7, cell[102]] = 1e12, np.inf, -np.inf   # overflow at p=4 / exact path
    # (whole-line chunking, fp32 (time, gridcell) applies: the scattered region is ~200 partial rows; region-shaped chunks,
    #  everything else: a multi-chunk "giant" group)
    from climate_toolbox_amd import _lib
    plan = SparsePlan(cell, code, w, G, Rn, row_len=nlon, flags=0 if lines else _lib.PLAN_NO_LINES)
    assert plan.info["lines"] == (7 if lines else 0) and plan.info["n_giant"] >= 1   # all three whole-line chunkings or none
    Xd = torch.from_numpy(X if layout == "TG" else np.ascontiguousarray(X.T)).cuda()
    got = plan.apply_poly(Xd, -273.15, 4, layout=layout).cpu().numpy()
    assert got.shape == (4, T, Rn)
    for p in range(1, 5):
        ref = O.agg_coded(O.tas_poly_values(X, p), cell, code, w, Rn)
        _rel_ok(got[p - 1], ref, RTOL32 if dtype == np.float32 else RTOL64, scale=1.0)
    # powers 3..8 (two fused passes that start above 1 where the loader/consumer kernel applies)
    Xs = np.where(np.isfinite(X), np.clip(X, 273.15 - 3, 273.15 + 3), X).astype(dtype)     # |y| <= 3: y^8 stays tame
    Xsd = torch.from_numpy(Xs if layout == "TG" else np.ascontiguousarray(Xs.T)).cuda()
    hi = plan.apply_poly(Xsd, -273.15, 6, layout=layout, pow_first=3).cpu().numpy()
    for i, p in enumerate(range(3, 9)):
        _rel_ok(hi[i], O.agg_coded(O.tas_poly_values(Xs, p), cell, code, w, Rn),
                RTOL32 if dtype == np.float32 else RTOL64, scale=1.0)
    # power 1 with offset 0 is the plain aggregation
    np.testing.assert_array_equal(plan.apply_poly(Xd, 0.0, 1, layout=layout)[0].cpu().numpy(),
                                  plan.apply(Xd, layout=layout).cpu().numpy())


def test_tas_poly_dropin_then_aggregate_and_batched(torch_cuda):
    """tas_poly(ds, p, name) -> weighted_aggregate_grid_to_regions(...) (the reference's two calls,
    transformations.py:160 + aggregations.py:87) and the one-pass tas_poly_aggregate give the
    oracle's transform-then-aggregate numbers; leap day removed, time relabelled to YYYYDDD."""
    from climate_toolbox_amd import (minixr, standardize_climate_data, tas_poly, tas_poly_aggregate,
                                     weighted_aggregate_grid_to_regions)
    from oracle import ref_numpy as O
    rng = np.random.default_rng(21)
    time = np.arange("2000-01-01", "2001-01-01", dtype="datetime64[D]")           # 366 days
    lat, lon = np.arange(-44.5, 45, 1.0), np.arange(0.5, 360, 1.0)                # raw 0..360 file
    tas = (288.15 + 8 * rng.standard_normal((len(time), len(lat), len(lon)))).astype(np.float32)
    tas[10, 5, 7] = np.nan
    ds = minixr.Dataset({"tas": (("time", "latitude", "longitude"), tas)},
                        coords={"time": time, "latitude": lat, "longitude": lon})
    ds = standardize_climate_data(ds)                                               # lazy lon wrap (8f-2)
    ref_grid, ref_lon = O.convert_lons_split(np.delete(tas, 59, axis=0), ("time", "lat", "lon"), lon)
    n = 4000
    df = pd.DataFrame({"lat": rng.choice(lat, n), "lon": rng.choice(ref_lon, n), "areawt": rng.uniform(0.1, 1, n),
                       "popwt": rng.uniform(0, 3, n), "hierid": rng.integers(0, 120, n)})
    refs = {}
    for p in (1, 2, 3, 4, 5):
        refs[p], rdims, labs = O.agg_scatter(O.tas_poly_values(ref_grid, p), ("time", "lat", "lon"), lat, ref_lon,
                                             df["lat"].values, df["lon"].values, df["popwt"].values,
                                             df["areawt"].values, df["hierid"].values, group_dim="hierid")
    # the reference's two-call form, one power at a time
    for p in (1, 3):
        out = weighted_aggregate_grid_to_regions(tas_poly(ds, p, "tas-poly-%d" % p), "tas-poly-%d" % p,
                                                 "popwt", "hierid", df)
        v = out["tas-poly-%d" % p]
        assert v.dims == ("time", "hierid") and v.shape == (365, len(labs))
        np.testing.assert_array_equal(out.time.values, 2000000 + np.arange(1, 366))
        _rel_ok(v.values, refs[p], RTOL32, scale=1.0)
    # all five powers, one pass for 1..4 + one for 5
    out = tas_poly_aggregate(ds, [1, 2, 3, 4, 5], "popwt", "hierid", df)
    assert list(out.data_vars) == ["tas-poly-%d" % p for p in (1, 2, 3, 4, 5)]
    for p in (1, 2, 3, 4, 5):
        _rel_ok(out["tas-poly-%d" % p].values, refs[p], RTOL32, scale=1.0)
    # lazy .values of the transformed grid itself (device-evaluated) equals the oracle's grid
    np.testing.assert_allclose(tas_poly(ds, 2, "p2").p2.values, O.tas_poly_values(ref_grid, 2), rtol=1e-6,
                               equal_nan=True)


def test_tile_sparse_dense_form_blocklocal(torch_cuda):
    """c5 "block-local" weights (each 64-cell run touches the 256 regions of one column tile): only
    the non-empty (32 x 256) tiles of W are stored and contracted by the MFMA kernel."""
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    G, R, T, seed = 64 * 75 + 40, 1000, 400, 5                    # ragged G and R, two row blocks
    W = O.blocklocal_weights_oracle(G, R, seed)
    plan = DensePlan.synth_blocklocal(G, R, seed)
    n_kt, n_nt = (G + 31) // 32, (R + 255) // 256
    assert plan.info["tiled"] == 1 and plan.info["n_tiles"] == n_kt and plan.info["n_tiles"] < n_kt * n_nt
    np.testing.assert_allclose(plan.den, W.astype(np.float64).sum(0), rtol=1e-12)
    rng = np.random.default_rng(2)
    X = (280 + 20 * rng.standard_normal((T, G))).astype(np.float32)
    X[7, 100] = np.nan
    ref = O.agg_dense(X, W)
    got = plan.apply(torch.from_numpy(X).cuda()).cpu().numpy()
    _rel_ok(got, ref, RTOL32)
    # the same weights handed over as a segment table take the tile-sparse form automatically ...
    gi, ri = np.nonzero(W)
    seg = DensePlan.from_segments(gi.astype(np.int32), ri.astype(np.int32), W[gi, ri].astype(np.float64), G, R)
    assert seg.info["tiled"] == 1 and seg.info["n_tiles"] == n_kt
    _rel_ok(seg.apply(torch.from_numpy(X).cuda()).cpu().numpy(), ref, RTOL32)
    # ... and agree with the fully dense layout of the same matrix
    full = DensePlan.from_host(W)
    assert full.info["tiled"] == 0 and full.info["n_tiles"] == n_kt * n_nt
    _rel_ok(full.apply(torch.from_numpy(X).cuda()).cpu().numpy(), ref, RTOL32)
    # empty column tiles / regions without any weight: 0/0
    gi2, ri2 = gi[ri < 300], ri[ri < 300]
    part = DensePlan.from_segments(gi2.astype(np.int32), ri2.astype(np.int32), W[gi2, ri2].astype(np.float64), G, R)
    out = part.apply(torch.from_numpy(X).cuda()).cpu().numpy()
    assert np.isnan(out[:, 300:]).all()
    _rel_ok(out[:, :300], ref[:, :300], RTOL32)


@pytest.mark.parametrize("dtype,layout", [(np.float32, "TG"), (np.float64, "TG"), (np.float32, "GT"), (np.float64, "GT")])
def test_fused_snyder_edd_matches_transform_then_aggregate(torch_cuda, dtype, layout):
    """Snyder exceedance degree days (transformations.py:7-93) of a (tasmin, tasmax)
    pair evaluated while both fields are loaded (wagg_apply_edd_*) = transform the grids, then
    aggregate; Kelvin inputs, thresholds in degrees C."""
    from climate_toolbox_amd import synth
    from climate_toolbox_amd.engine import SparsePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    nlat, nlon, R, T = 96, 192, 300, 70
    lat, lon, df = synth.realistic_segments(nlat, nlon, R=R, seed=6, string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
    rng = np.random.default_rng(9)
    extra = rng.choice(nlat * nlon, 700, replace=False).astype(np.int32)        # one giant region
    cell = np.concatenate([cell, extra]); code = np.concatenate([code, np.full(700, len(uniq), np.int32)])
    w = np.concatenate([w, rng.uniform(0.1, 1, 700)])
    Rn, G = len(uniq) + 1, nlat * nlon
    tmean = 273.15 + 22 + 8 * rng.standard_normal((T, G))
    half = rng.uniform(0, 8, (T, G))
    tmin, tmax = (tmean - half).astype(dtype), (tmean + half).astype(dtype)
    tmin[2, cell[:30]] = np.nan                      # NaN tasmin -> NaN EDD -> skipped product (S6)
    tmax[3, cell[30:60]] = np.nan                    # NaN tasmax: 0 where tasmin < e, NaN elsewhere
    tmax[4, cell[60:90]] = tmin[4, cell[60:90]]      # zero width
    thr = [10.0, 25.0, 30.0]
    plan = SparsePlan(cell, code, w, G, Rn, row_len=nlon)
    a, b = (tmin, tmax) if layout == "TG" else (np.ascontiguousarray(tmin.T), np.ascontiguousarray(tmax.T))
    got = plan.apply_edd(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), thr, offset=-273.15,
                         layout=layout).cpu().numpy()
    assert got.shape == (3, T, Rn)
    ft = np.dtype(dtype).type
    cmin, cmax = tmin + ft(-273.15), tmax + ft(-273.15)
    refs = [O.agg_coded(O.snyder_edd_values(cmin, cmax, e), cell, code, w, Rn) for e in thr]
    for i in range(3):
        _rel_ok(got[i], refs[i], RTOL32 if dtype == np.float32 else RTOL64, scale=0.05)
    # growing degree days = difference of two aggregated EDDs (transformations.py:138-140; linear)
    gdd_ref = O.agg_coded(O.snyder_gdd_values(cmin, cmax, 10.0, 30.0), cell, code, w, Rn)
    _rel_ok(got[0] - got[2], gdd_ref, RTOL32 if dtype == np.float32 else RTOL64, scale=0.05)


def test_snyder_degree_days_dropin(torch_cuda):
    """snyder_edd / snyder_gdd (transformations.py:7-144) as lazy variables of a Dataset, aggregated
    by the reference's own entry point; Kelvin fields via convert_kelvin_to_celsius."""
    from climate_toolbox_amd import minixr, snyder_edd, snyder_gdd, weighted_aggregate_grid_to_regions
    from climate_toolbox_amd.transformations import convert_kelvin_to_celsius
    from oracle import ref_numpy as O
    rng = np.random.default_rng(31)
    lat, lon = np.arange(-29.5, 30, 1.0), np.arange(-59.5, 60, 1.0)
    T = 45
    mean = 273.15 + 24 + 7 * rng.standard_normal((T, len(lat), len(lon)))
    half = rng.uniform(0, 9, mean.shape)
    tmin, tmax = (mean - half).astype(np.float32), (mean + half).astype(np.float32)
    tmin[5, 3, 4] = np.nan
    ds = minixr.Dataset({"tasmin": (("time", "lat", "lon"), tmin), "tasmax": (("time", "lat", "lon"), tmax)},
                        coords={"time": np.arange(T), "lat": lat, "lon": lon})
    for name in ("tasmin", "tasmax"):
        ds[name].attrs["units"] = "K"
        ds = convert_kelvin_to_celsius(ds, name)                      # lazy offset (utils.py:10-20)
    assert ds.tasmin.attrs["units"] == "C"
    n = 2500
    df = pd.DataFrame({"lat": rng.choice(lat, n), "lon": rng.choice(lon, n), "areawt": rng.uniform(0.1, 1, n),
                       "popwt": rng.uniform(0, 2, n), "ISO": rng.integers(0, 40, n)})
    cmin, cmax = tmin + np.float32(-273.15), tmax + np.float32(-273.15)
    ds["edd30"] = snyder_edd(ds.tasmin, ds.tasmax, 30)
    ds["gdd"] = snyder_gdd(ds.tasmin, ds.tasmax, 10, 30)
    assert ds["edd30"].attrs["units"] == "degreedays_30C" and ds["gdd"].attrs["units"] == "degreedays_10-30C"
    for var, grid in (("edd30", O.snyder_edd_values(cmin, cmax, 30)), ("gdd", O.snyder_gdd_values(cmin, cmax, 10, 30))):
        ref, rdims, labs = O.agg_scatter(grid, ("time", "lat", "lon"), lat, lon, df["lat"].values, df["lon"].values,
                                         df["popwt"].values, df["areawt"].values, df["ISO"].values, group_dim="ISO")
        out = weighted_aggregate_grid_to_regions(ds, var, "popwt", "ISO", df)
        assert out[var].dims == rdims and list(out["ISO"].values) == list(labs)
        _rel_ok(out[var].values, ref, RTOL32, scale=0.05)
        np.testing.assert_allclose(ds[var].values, grid, rtol=2e-5, atol=2e-5, equal_nan=True)   # lazy .values
    # tasmin > tasmax is refused like in the reference (transformations.py:62)
    bad = minixr.Dataset({"tasmin": (("time", "lat", "lon"), tmax), "tasmax": (("time", "lat", "lon"), tmin)},
                         coords={"time": np.arange(T), "lat": lat, "lon": lon})
    with pytest.raises(AssertionError):
        snyder_edd(bad.tasmin, bad.tasmax, 30)


def test_randomised_differential_all_entry_points(torch_cuda):
    """tests/fuzz_gpu.py, 80 seeded cases: random grids, tables (null labels, NaN / 0 weights,
    giant regions), layouts, padded row strides, NaN / inf data -- plain, fused-power, degree-day
    and dense / tile-sparse forms against the oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fuzz_gpu", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["fuzz_gpu.py"]
    try:
        spec.loader.exec_module(fz)
    finally:
        sys.argv = argv
    rng = np.random.default_rng(2024)
    failures = []
    for i in range(80):
        tag, fails = fz.one_case(i, rng)
        if fails:
            failures.append(tag + " | " + "; ".join(fails))
    assert not failures, "\n".join(failures)


def test_single_column_view_of_a_padded_buffer(torch_cuda):
    """(G, 1) data with a row pitch > 1 (found by the fuzzer): the leading dimension is the pitch."""
    from climate_toolbox_amd.engine import SparsePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(3)
    G, R = 500, 20
    cell, code = rng.integers(0, G, 900).astype(np.int32), rng.integers(0, R, 900).astype(np.int32)
    w = rng.uniform(0.1, 1, 900)
    X = rng.standard_normal((1, G))
    buf = torch.zeros((G, 4), dtype=torch.float64, device="cuda")
    buf[:, 0] = torch.from_numpy(X[0]).cuda()
    got = SparsePlan(cell, code, w, G, R).apply(buf[:, :1], layout="GT").cpu().numpy()
    _rel_ok(got, O.agg_coded(X, cell, code, w, R), RTOL64)


@pytest.mark.parametrize("dims", [("time", "lat", "lon"), ("lat", "lon", "time"), ("lat", "time", "lon"),
                                  ("member", "time", "lat", "lon"), ("lat", "lon", "member", "time"),
                                  ("time", "lon", "lat"), ("lon", "member", "lat"), ("lat", "lon")])
def test_dropin_any_dim_order_and_label_types(torch_cuda, dims):
    """S2/S3/S10 through the drop-in: lat/lon anywhere among the dims (also lon before lat), extra
    dims carried in place, string labels with nulls, against the oracle's xarray-semantics restatement."""
    from climate_toolbox_amd import minixr, weighted_aggregate_grid_to_regions
    from oracle import ref_numpy as O
    rng = np.random.default_rng(len(dims) * 7 + len(dims[0]))
    sizes = {"time": 9, "member": 3, "lat": 12, "lon": 17}
    lat, lon = np.linspace(-55, 55, sizes["lat"]), np.linspace(-160, 170, sizes["lon"])
    vals = (280 + 10 * rng.standard_normal([sizes[d] for d in dims]))
    vals.flat[rng.integers(0, vals.size, 3)] = np.nan
    coords = {d: (lat if d == "lat" else lon if d == "lon" else np.arange(sizes[d])) for d in dims}
    ds = minixr.Dataset({"tas": (dims, vals)}, coords=coords)
    n = 400
    labels = np.array(["R%02d" % k for k in rng.integers(0, 25, n)], dtype=object)
    labels[rng.random(n) < 0.05] = None                                  # null labels are dropped (S3)
    df = pd.DataFrame({"lat": rng.choice(lat, n), "lon": rng.choice(lon, n), "areawt": rng.uniform(0.1, 1, n),
                       "popwt": np.where(rng.random(n) < 0.2, np.nan, rng.uniform(0, 4, n)), "hierid": labels})
    ref, rdims, labs = O.agg_scatter(vals, dims, lat, lon, df["lat"].values, df["lon"].values, df["popwt"].values,
                                     df["areawt"].values, df["hierid"].values, group_dim="hierid")
    out = weighted_aggregate_grid_to_regions(ds, "tas", "popwt", "hierid", df)
    assert out.tas.dims == rdims and list(out["hierid"].values) == list(labs)
    for d in rdims:
        if d != "hierid":
            np.testing.assert_array_equal(out[d].values, coords[d])
    _rel_ok(out.tas.values, ref, RTOL64)


def test_dropin_accepts_device_resident_fields(torch_cuda):
    """A variable whose buffer is a torch CUDA tensor is aggregated in place (no PCIe copy) -- also
    through the lazy standardize / tas_poly wrappers and for a layout that needs a transpose."""
    from climate_toolbox_amd import minixr, standardize_climate_data, tas_poly, weighted_aggregate_grid_to_regions
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(77)
    lat, lon = np.arange(-29.5, 30, 1.0), np.arange(0.5, 120, 1.0)
    time = np.arange("2000-02-01", "2000-04-01", dtype="datetime64[D]")        # holds a 29 February
    tas = (288 + 6 * rng.standard_normal((len(time), len(lat), len(lon)))).astype(np.float32)
    leap = int(np.flatnonzero(time == np.datetime64("2000-02-29"))[0])
    n = 1500
    for dims, order in ((("time", "lat", "lon"), (0, 1, 2)), (("lat", "time", "lon"), (1, 0, 2))):
        dev = torch.from_numpy(np.ascontiguousarray(tas.transpose(order))).cuda()
        ds = minixr.Dataset({"tas": (dims, dev)}, coords={"time": time, "lat": lat, "lon": lon})
        ds = standardize_climate_data(ds)
        ref_grid, ref_lon = O.convert_lons_split(np.delete(tas, leap, axis=0), ("time", "lat", "lon"), lon)
        df = pd.DataFrame({"lat": rng.choice(lat, n), "lon": rng.choice(ref_lon, n), "areawt": rng.uniform(0.1, 1, n),
                           "hierid": rng.integers(0, 60, n)})
        ref, _, labs = O.agg_scatter(O.tas_poly_values(ref_grid, 2), ("time", "lat", "lon"), lat, ref_lon,
                                     df["lat"].values, df["lon"].values, df["areawt"].values, df["areawt"].values,
                                     df["hierid"].values, group_dim="hierid")
        out = weighted_aggregate_grid_to_regions(tas_poly(ds, 2, "t2"), "t2", "areawt", "hierid", df)
        got = out.t2.values if dims[0] == "time" else np.moveaxis(out.t2.values, 0, 1)
        _rel_ok(got, ref, RTOL32, scale=1.0)


def test_integration_md_binding_runs_as_written(torch_cuda):
    """The ctypes stub INTEGRATION.md proposes for the reference (section 2) is executed verbatim
    against libwagg.so and checked against the oracle."""
    import re
    from climate_toolbox_amd import _lib
    from oracle import ref_numpy as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# climate_toolbox/aggregations/_wagg.py.*?)```", text, re.S).group(1)
    block = block.replace('C.CDLL("libwagg.so")', "C.CDLL(%r)" % _lib.LIB_PATH)
    _lib.load()                                             # torch's HIP runtime first, like the product
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    rng = np.random.default_rng(12)
    lat, lon = np.arange(-10.0, 10.0, 1.0), np.arange(100.0, 130.0, 1.0)
    vals = rng.standard_normal((6, len(lat) * len(lon)))
    n = 300
    df = pd.DataFrame({"lat": rng.choice(lat, n), "lon": rng.choice(lon, n), "areawt": rng.uniform(0.1, 1, n),
                       "popwt": np.where(rng.random(n) < 0.3, np.nan, rng.uniform(0, 2, n)),
                       "ISO": rng.integers(0, 9, n)})
    out, labels = ns["aggregate"](np.ascontiguousarray(vals), lat, lon, df, "popwt", "ISO")
    ref, _, labs = O.agg_scatter(vals.reshape(6, len(lat), len(lon)), ("time", "lat", "lon"), lat, lon,
                                 df["lat"].values, df["lon"].values, df["popwt"].values, df["areawt"].values,
                                 df["ISO"].values, group_dim="ISO")
    assert list(labels) == list(labs)
    _rel_ok(out, ref, RTOL64)


def test_integration_md_descriptor_binding_runs_as_written(torch_cuda):
    """The descriptor form INTEGRATION.md shows beside the stub (wagg_apply + a hand-written mirror of wagg_apply_desc) is
    executed as written, in the stub's namespace: a plain host apply and a fused-powers one, against the oracle."""
    import ctypes as C
    import re
    from climate_toolbox_amd import _lib, synth
    from climate_toolbox_amd.engine import SparsePlan
    from oracle import ref_numpy as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    base = re.search(r"```python\n(# climate_toolbox/aggregations/_wagg.py.*?)```", text, re.S).group(1)
    base = base.replace('C.CDLL("libwagg.so")', "C.CDLL(%r)" % _lib.LIB_PATH)
    block = re.search(r"```python\n(class ApplyDesc\(C.Structure\):.*?)```", text, re.S).group(1)
    _lib.load()
    ns = {}
    exec(compile(base, "INTEGRATION.md", "exec"), ns)
    exec(compile(block, "INTEGRATION.md#descriptor", "exec"), ns)
    assert [f for f, _ in ns["ApplyDesc"]._fields_] == [f for f, _ in _lib.ApplyDesc._fields_]
    lat, lon, df = synth.realistic_segments(nlat=96, nlon=192, R=150, n_iso=10, seed=3, land_frac=0.2, string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
    G, R, T = len(lat) * len(lon), len(uniq), 37
    plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
    rng = np.random.default_rng(4)
    for dtype, rtol in ((np.float32, RTOL32), (np.float64, RTOL64)):
        X = (273.15 + 25 * rng.random((T, G))).astype(dtype)
        out = ns["apply_host_desc"](plan._h, X, R)
        _rel_ok(out, O.agg_coded(X.astype(np.float64), cell, code, w, R), rtol)
        stack = ns["apply_host_desc"](plan._h, X, R, powers=(2, 3), offset=-273.15)
        assert stack.shape == (3, T, R)
        for i, pw in enumerate((2, 3, 4)):
            _rel_ok(stack[i], O.agg_coded(O.tas_poly_values(X.astype(np.float64), pw), cell, code, w, R), rtol * 4, scale=1.0)
    plan.close()


def test_c_abi_argument_errors_are_reported_not_crashed(torch_cuda):
    """Every entry point returns a negative status with a message for bad arguments (no throw, no
    exit, no launch): shapes, strides, ranges."""
    import ctypes as C
    from climate_toolbox_amd import _lib
    from climate_toolbox_amd.engine import SparsePlan, DensePlan
    torch = torch_cuda
    L = _lib.load()
    G, R = 100, 5
    plan = SparsePlan(np.arange(50, dtype=np.int32), (np.arange(50) % R).astype(np.int32), np.ones(50), G, R)
    X = torch.ones((4, G), dtype=torch.float32, device="cuda")
    out = torch.empty((3, 4, R), dtype=torch.float32, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())
    def err(rc):
        assert rc < 0
        return L.wagg_last_error().decode()
    assert "ldx" in err(L.wagg_apply_f32(plan._h, vp(X), 4, G - 1, 0, vp(out), R, 0, None))
    assert "layout" in err(L.wagg_apply_f32(plan._h, vp(X), 4, G, 7, vp(out), R, 0, None))
    assert "NULL" in err(L.wagg_apply_f32(None, vp(X), 4, G, 0, vp(out), R, 0, None)).upper()
    assert "powers" in err(L.wagg_apply_poly_f32(plan._h, vp(X), 4, G, 0, 0.0, 1, 17, vp(out), R, 4 * R, 0, None))
    assert "powers" in err(L.wagg_apply_poly_f32(plan._h, vp(X), 4, G, 0, 0.0, 0, 2, vp(out), R, 4 * R, 0, None))
    assert "overlaps" in err(L.wagg_apply_poly_f32(plan._h, vp(X), 4, G, 0, 0.0, 1, 3, vp(out), R, R, 0, None))
    thr = (C.c_double * 2)(10.0, 20.0)
    assert "tasmax" in err(L.wagg_apply_edd_f32(plan._h, vp(X), None, 4, G, 0, 0.0, thr, 2, vp(out), R, 4 * R, 0, None))
    assert "threshold" in err(L.wagg_apply_edd_f32(plan._h, vp(X), vp(X), 4, G, 0, 0.0, thr, 0, vp(out), R, 4 * R, 0, None))
    h = C.c_void_p()
    assert err(L.wagg_plan_create(None, None, None, 5, G, R, 0, 0, C.byref(h))) and not h.value
    assert err(L.wagg_dense_create_synth(0, R, 1, C.byref(h))) and not h.value
    dense = DensePlan.synth(64, 8, 1)
    assert "ksplit" in err(L.wagg_dense_apply_f32(dense._h, vp(X), 4, 64, vp(out), 8, 3, None))
    assert "ldx" in err(L.wagg_dense_apply_f32(dense._h, vp(X), 4, 63, vp(out), 8, 0, None))
    # the plans are still usable afterwards
    assert torch.isfinite(plan.apply(X)).all()


def test_concurrent_applies_on_two_streams_share_one_plan(torch_cuda):
    """include/wagg.h: a sparse plan is immutable, applies on distinct streams may overlap.  Two
    streams aggregate different fields through one plan at the same time, many times over."""
    from climate_toolbox_amd import synth
    from climate_toolbox_amd.engine import SparsePlan
    torch = torch_cuda
    lat, lon, df = synth.realistic_segments(180, 360, R=2000, seed=3, string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
    G, R, T = len(lat) * len(lon), len(uniq), 200
    plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
    gen = torch.Generator(device="cuda").manual_seed(1)
    XA = 280 + 20 * torch.randn((T, G), device="cuda", generator=gen)
    XB = 260 + 30 * torch.randn((T, G), device="cuda", generator=gen)
    refA, refB = plan.apply(XA).clone(), plan.apply(XB).clone()
    refP = plan.apply_poly(XB, -273.15, 3).clone()
    torch.cuda.synchronize()
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    outsA, outsB, outsP = [], [], []
    for _ in range(10):
        with torch.cuda.stream(sA):
            outsA.append(plan.apply(XA))
        with torch.cuda.stream(sB):
            outsB.append(plan.apply(XB))
            outsP.append(plan.apply_poly(XB, -273.15, 3))
    torch.cuda.synchronize()
    for o in outsA:
        assert torch.equal(o, refA)
    for o in outsB:
        assert torch.equal(o, refB)
    for o in outsP:
        assert torch.equal(o, refP)
