"""GPU parity tests added in round 2 (run with -m gpu on an MI355X): the numbers the reference's own
tests hold, the weights-CSV path, the entry-list form for scattered weights (c5 "uniform"), c5 at
its full rank-shard size, transforms in the dense-family pack stage, +-inf routing, and the
device-failure status.  Every comparison goes through the C-ABI; the oracle is only the checker."""
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from tests.test_gpu_parity import RTOL32, RTOL64, _rel_ok, _tile_windows

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


# ---------------------------------------------------------------------------------------------
# numbers the reference's own tests hold (tests/golden/reference_held.npz)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_reference_held_snyder_vectors_lazy_and_fused(torch_cuda, golden_dir, dtype):
    """tests/test_climate_toolbox.py:242-290 through BOTH paths: the lazy variable's ``.values``
    (wagg_transform_edd_*) and the fused aggregation (wagg_apply_edd_*) over a one-region plan that
    covers the two cells: snyder_edd sums to exactly 0.0, snyder_gdd to ~11, units strings equal."""
    from climate_toolbox_amd import minixr, snyder_edd, snyder_gdd, weighted_aggregate_grid_to_regions
    from oracle import ref_numpy as O
    h = np.load(os.path.join(golden_dir, "reference_held.npz"), allow_pickle=False)
    lat, lon = np.array([-33.625]), np.array([286.125, 286.375])                 # the test's coordinates
    tmin = h["edd_tmin"].astype(dtype).reshape(1, 1, 2)
    tmax = h["edd_tmax"].astype(dtype).reshape(1, 1, 2)
    ds = minixr.Dataset({"tmin": (("time", "lat", "lon"), tmin), "tmax": (("time", "lat", "lon"), tmax)},
                        coords={"time": np.arange(1), "lat": lat, "lon": lon})
    ds["tmin"].attrs["units"] = ds["tmax"].attrs["units"] = str(h["edd_units_in"])
    thr, lo, hi = float(h["edd_threshold"]), float(h["gdd_threshold_low"]), float(h["gdd_threshold_high"])
    edd = snyder_edd(ds.tmin, ds.tmax, threshold=thr)
    gdd = snyder_gdd(ds.tmin, ds.tmax, threshold_low=lo, threshold_high=hi)
    assert edd.attrs["units"] == str(h["edd_units"])                              # :261
    assert gdd.attrs["units"] == str(h["gdd_units"]) and gdd.attrs["units"] != str(h["edd_units"])   # :288-289
    # lazy .values (device transform kernel)
    assert edd.values.sum() == float(h["edd_sum"]) == 0.0                         # :262
    assert gdd.values.sum() == pytest.approx(float(h["gdd_sum_approx"]), float(h["gdd_sum_rel"]))   # :290
    np.testing.assert_allclose(gdd.values, O.snyder_gdd_values(tmin, tmax, lo, hi), rtol=1e-6)
    # fused: one region owning both cells with unit weights -> mean = sum / 2
    ds["edd"], ds["gdd"] = edd, gdd
    df = pd.DataFrame({"lat": [lat[0], lat[0]], "lon": lon, "areawt": [1.0, 1.0], "hierid": ["r", "r"]})
    out_e = weighted_aggregate_grid_to_regions(ds, "edd", "areawt", "hierid", df)
    out_g = weighted_aggregate_grid_to_regions(ds, "gdd", "areawt", "hierid", df)
    assert out_e.edd.values.shape == (1, 1) and out_e.edd.values[0, 0] * 2 == 0.0
    assert out_g.gdd.values[0, 0] * 2 == pytest.approx(float(h["gdd_sum_approx"]), float(h["gdd_sum_rel"]))
    np.testing.assert_allclose(out_g.gdd.values[0, 0] * 2, O.snyder_gdd_values(tmin, tmax, lo, hi).sum(), rtol=1e-6)


def test_reference_held_convert_lons_then_aggregate(torch_cuda, golden_dir):
    """tests/test_climate_toolbox.py:180-197 on a dataset that carries data: the relabelled, sorted
    longitudes equal the held vectors and the lazily re-ordered field aggregates like the eagerly
    re-ordered one."""
    from climate_toolbox_amd import convert_lons_mono, convert_lons_split, minixr, weighted_aggregate_grid_to_regions
    from oracle import ref_numpy as O
    h = np.load(os.path.join(golden_dir, "reference_held.npz"), allow_pickle=False)
    rng = np.random.default_rng(0)
    for fn, oracle_fn, lon_in, expect, name in (
            (convert_lons_mono, O.convert_lons_mono, h["mono_lon"], h["mono_expect"], "lon"),
            (convert_lons_split, O.convert_lons_split, h["split_lon"].astype(np.float64), h["split_expect"].astype(np.float64), "lon")):
        lat = np.array([10.0, 20.0, 30.0])
        tas = rng.standard_normal((4, 3, 2))
        ds = fn(minixr.Dataset({"tas": (("time", "lat", "lon"), tas)}, coords={"time": np.arange(4), "lat": lat, name: lon_in}),
                lon_name=name)
        np.testing.assert_array_equal(ds.lon.values, expect)
        vals, labs = oracle_fn(tas, ("time", "lat", "lon"), lon_in)
        np.testing.assert_array_equal(ds.tas.values, vals)
        df = pd.DataFrame({"lat": lat[[0, 1, 2, 1]], "lon": expect[[0, 1, 1, 0]], "areawt": [1.0, 2.0, 3.0, 4.0],
                           "hierid": [1, 1, 2, 2]})
        out = weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
        ref, _, _ = O.agg_scatter(vals, ("time", "lat", "lon"), lat, labs, df["lat"].values, df["lon"].values,
                                  df["areawt"].values, df["areawt"].values, df["hierid"].values, group_dim="hierid")
        _rel_ok(out.tas.values, ref, RTOL64)


# ---------------------------------------------------------------------------------------------
# prepare_spatial_weights_data through the drop-in (aggregations.py:127-152 -> :118-124)
# ---------------------------------------------------------------------------------------------
def test_weights_csv_path_with_dateline_pixel(torch_cuda, tmp_path):
    """``weights`` given as the path of the CSV (aggregations.py:108): pix_cent_x/pix_cent_y columns,
    a pixel centre at 180.125 that must land on the grid's -179.875 column (:144), duplicate rows
    kept (the reference discards drop_duplicates' result, :147)."""
    from climate_toolbox_amd import minixr, prepare_spatial_weights_data, weighted_aggregate_grid_to_regions
    from oracle import ref_numpy as O
    lat = -89.875 + 0.25 * np.arange(720)[300:340]
    lon = -179.875 + 0.25 * np.arange(1440)[:48]
    rng = np.random.default_rng(5)
    tas = (280 + 10 * rng.standard_normal((6, len(lat), len(lon)))).astype(np.float32)
    rows = [(180.125, lat[3], 0.7, 1.5, "USA.1", "USA"), (-179.625, lat[3], 0.2, np.nan, "USA.1", "USA"),
            (180.125, lat[4], 0.5, 0.0, "RUS.9", "RUS"), (-170.125, lat[20], 0.9, 2.0, "RUS.9", "RUS"),
            (-170.125, lat[20], 0.9, 2.0, "RUS.9", "RUS"), (-168.375, lat[39], 0.4, 0.3, "FJI.2", "FJI")]
    csv = tmp_path / "segment_weights.csv"
    pd.DataFrame(rows, columns=["pix_cent_x", "pix_cent_y", "areawt", "popwt", "hierid", "ISO"]).to_csv(csv, index=False)
    ds = minixr.Dataset({"tas": (("time", "lat", "lon"), tas)}, coords={"time": np.arange(6), "lat": lat, "lon": lon})
    df = prepare_spatial_weights_data(str(csv))
    assert list(df.columns[:2]) == ["lon", "lat"] and df.index.names == ["reshape_index"]
    assert (df["lon"].values[[0, 2]] == -179.875).all() and len(df) == 6
    for aggwt, agglev in (("areawt", "hierid"), ("popwt", "ISO")):
        out = weighted_aggregate_grid_to_regions(ds, "tas", aggwt, agglev, weights=str(csv))
        seg_lon = np.where(np.array([r[0] for r in rows]) == 180.125, -179.875, [r[0] for r in rows])
        ref, rdims, labs = O.agg_scatter(tas, ("time", "lat", "lon"), lat, lon, np.array([r[1] for r in rows]), seg_lon,
                                         np.array([r[3] if aggwt == "popwt" else r[2] for r in rows], dtype=float),
                                         np.array([r[2] for r in rows]), np.array([r[4] if agglev == "hierid" else r[5] for r in rows], dtype=object),
                                         group_dim=agglev)
        assert out.tas.dims == rdims and list(out[agglev].values) == list(labs)
        _rel_ok(out.tas.values, ref, RTOL32)
    # a pixel that is not on the grid raises KeyError like Dataset.sel (S1)
    bad = tmp_path / "bad.csv"
    pd.DataFrame([(0.125, lat[0], 1.0, 1.0, "X", "X")], columns=["pix_cent_x", "pix_cent_y", "areawt", "popwt", "hierid", "ISO"]).to_csv(bad, index=False)
    with pytest.raises(KeyError):
        weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", weights=str(bad))


# ---------------------------------------------------------------------------------------------
# entry-list form (wagg_spmm.hip): scattered, sparse weights
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,rtol", [(np.float32, RTOL32), (np.float64, RTOL64)])
@pytest.mark.parametrize("T,G,R,fill", [(185, 64 * 96, 300, 0.01), (64, 256, 17, 0.05), (1, 300, 5, 0.09),
                                        (130, 5000, 1600, 0.02), (400, 1537, 96, 0.03), (77, 700, 700, 0.095)])
def test_entry_list_form_synth_vs_oracle(torch_cuda, T, G, R, fill, dtype, rtol):
    """wagg_dense_create_synth_sparse / _f64 below 10 % fill build entry lists on the device; every
    region-timestep against the fp64 oracle on the regenerated matrix (ragged T, G, R; two region
    blocks at R = 1600; a list of more than 16 groups -- the slow tail of the generated loop -- at
    700 cells x 700 regions x 9.5 %: ~270 entries per wave and chunk).  fp64 (the reference's own arithmetic type,
    aggregations.py:73-80) at 1e-6, measured ~1e-13."""
    from climate_toolbox_amd import _lib
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    seed = 11
    W = O.dense_weights_oracle(G, R, seed, fill)
    plan = DensePlan.synth(G, R, seed, fill=fill, dtype=dtype)
    assert plan.info["form"] == _lib.FORM_ENTRIES and plan.info["nnz"] == int((W != 0).sum())
    assert plan.dtype == np.dtype(dtype).name
    np.testing.assert_allclose(plan.den, W.astype(np.float64).sum(0), rtol=1e-12)
    rng = np.random.default_rng(3)
    X = (280 + 20 * rng.standard_normal((T, G))).astype(dtype)
    X[0, G // 2] = np.nan
    ref = O.agg_dense(X, W)
    Xd = torch.from_numpy(X).cuda()
    got = plan.apply(Xd).cpu().numpy()
    _rel_ok(got, ref, rtol)
    if dtype == np.float64:
        _rel_ok(got, ref, 1e-11)                                                   # (what fp64 accumulation really gives)
    np.testing.assert_array_equal(plan.apply(Xd).cpu().numpy(), got)              # bitwise reproducible
    # the same weights handed over as a segment table take the same form and give the same numbers
    gi, ri = np.nonzero(W)
    seg = DensePlan.from_segments(gi.astype(np.int32), ri.astype(np.int32), W[gi, ri].astype(np.float64), G, R, dtype=dtype)
    if len(gi) < 0.1 * G * R and seg.info["tiled"] == 0:
        assert seg.info["form"] == _lib.FORM_ENTRIES
        np.testing.assert_allclose(seg.apply(Xd).cpu().numpy(), got, rtol=1e-6 if dtype == np.float32 else 1e-12, equal_nan=True)
    # transforms in the pack stage: (x - 273.15)^2 and one degree-day threshold
    _rel_ok(plan.apply_poly(Xd, -273.15, 2).cpu().numpy(), O.agg_dense(O.tas_poly_values(X, 2), W), rtol)
    Xhi = X + dtype(6.0)
    edd = O.snyder_edd_values(X + dtype(-273.15), Xhi + dtype(-273.15), 10.0)
    _rel_ok(plan.apply_edd(Xd, torch.from_numpy(Xhi).cuda(), 10.0, offset=-273.15).cpu().numpy(),
            O.agg_dense(edd, W), rtol, scale=0.05)
    assert not plan.saw_inf()


@pytest.mark.parametrize("dtype,rtol", [(np.float32, RTOL32), (np.float64, 1e-11)])
def test_entry_list_form_keeps_inf_with_the_owning_regions(torch_cuda, dtype, rtol):
    """+-inf data (S6): the entry-list form multiplies real (cell, region) pairs only, so an inf
    reaches exactly the regions that own the cell -- like the reference and the segment-table form."""
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(8)
    T, G, R = 70, 2048, 200
    nnz = int(0.02 * G * R)
    flat = rng.choice(G * R, size=nnz, replace=False)
    cell, code = (flat // R).astype(np.int32), (flat % R).astype(np.int32)
    w = rng.uniform(0.1, 1.0, nnz)
    X = (280 + 15 * rng.standard_normal((T, G))).astype(dtype)
    X[3, 17] = np.inf
    X[5, 900] = -np.inf
    X[6, 901] = np.nan
    ref = O.agg_coded(X, cell, code, w, R)
    plan = DensePlan.from_segments(cell, code, w, G, R, dtype=dtype)
    assert plan.info["form"] == 2
    got = plan.apply(torch.from_numpy(X).cuda()).cpu().numpy()
    assert np.isinf(ref).sum() > 0 and np.isinf(ref).sum() < 0.1 * ref.size
    _rel_ok(got, ref, rtol)


def test_dropin_routes_inf_around_the_mfma_forms(torch_cuda):
    """the drop-in picks the dense MFMA form for scattered weights by itself; a field with
    +-inf (also one that only overflows after the fused power) must still come out as in the
    reference -- inf confined to the regions that own the cell -- not as NaN everywhere."""
    from climate_toolbox_amd import aggregations as A, minixr, tas_poly_aggregate
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    rng = np.random.default_rng(3)
    nlat, nlon, R, T = 24, 48, 1024, 30                         # (four full column tiles of regions: with 40 regions the form
    lat, lon = np.arange(nlat) * 1.0, np.arange(nlon) * 1.0    #  choice of round 5 rightly prefers entry lists to one tile that is
                                                                #  84 % padding and fills a quarter of the CUs)
    n = int(0.3 * nlat * nlon * R)                              # dense-ish random table -> an MFMA form (full matrix or all tiles stored)
    flat = rng.choice(nlat * nlon * R, size=n, replace=False)
    cell, lab = flat // R, flat % R
    df = pd.DataFrame({"lat": lat[cell // nlon], "lon": lon[cell % nlon], "areawt": rng.uniform(0.1, 1, n), "hierid": lab})
    tas = (280 + 10 * rng.standard_normal((T, nlat, nlon))).astype(np.float32)
    tas[2, 5, 7] = np.inf
    tas[4, 9, 1] = -np.inf
    tas[4, 9, 2] = np.nan
    ds = minixr.Dataset({"tas": (("time", "lat", "lon"), tas)}, coords={"time": pd.date_range("2001-01-01", periods=T).values, "lat": lat, "lon": lon})
    A._PLAN_CACHE.clear()
    out = A.weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
    assert any(isinstance(p, DensePlan) and p.info["form"] in (0, 1) for p in A._PLAN_CACHE.values())     # an MFMA form (full or tiles)
    args = (("time", "lat", "lon"), lat, lon, df["lat"].values, df["lon"].values, df["areawt"].values, df["areawt"].values, df["hierid"].values)
    ref = O.agg_scatter(tas, *args, group_dim="hierid")[0]
    assert np.isinf(ref).any() and np.isfinite(ref).any()
    _rel_ok(out.tas.values, ref, RTOL32)
    # overflow only after the transform: 1e13 ** 3 is inf in fp32
    tas2 = tas.copy()
    tas2[2, 5, 7] = 1.0e13
    tas2[4, 9, 1] = 280.0
    ds2 = minixr.Dataset({"tas": (("time", "lat", "lon"), tas2)}, coords=ds.coords)
    outp = tas_poly_aggregate(ds2, [1, 3], "areawt", "hierid", df)
    for p in (1, 3):
        refp = O.agg_scatter(O.tas_poly_values(tas2, p), *args, group_dim="hierid")[0]
        _rel_ok(outp["tas-poly-%d" % p].values, refp, RTOL32, scale=1e-3 if p == 1 else 1.0)


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[4] at its full rank-shard size (T = 2,282 of 18,250 rows, G = 1,036,800,
# R = 24,378, ~2.53e8 non-zeros), both structures of SURVEY 8d
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("structure,dtype", [("uniform", "float32"), ("block_local", "float32"), ("uniform", "float64")])
def test_c5_full_size_rank_shard(torch_cuda, structure, dtype):
    from climate_toolbox_amd import _lib, engine
    from climate_toolbox_amd.timeshard import shard_bounds
    from oracle import c_oracle, ref_numpy as O
    torch = torch_cuda
    G, R, seed = 720 * 1440, 24378, 2
    bounds = shard_bounds(50 * 365, 8)
    T = bounds[0][1] - bounds[0][0]
    assert T == 2282 and bounds[-1][1] - bounds[-1][0] == 2281
    uniform = structure == "uniform"
    fill = 0.01 if uniform else 0.952
    f64 = dtype == "float64"
    plan = (engine.DensePlan.synth(G, R, seed, fill=fill, dtype=dtype) if uniform
            else engine.DensePlan.synth_blocklocal(G, R, seed, fill=fill, dtype=dtype))
    assert plan.info["form"] == (_lib.FORM_ENTRIES if uniform else _lib.FORM_TILES) and plan.dtype == dtype
    if uniform:
        assert abs(plan.info["nnz"] - 0.01 * G * R) < 2e-4 * G * R         # ~2.53e8 kept pairs
    X = engine.synth_field(T, G, seed=1000, base=280.0, amp=60.0)          # fp32 values (the C oracle takes fp32 fields) ...
    if f64:
        X = X.double()                                                      # ... held in fp64 for the fp64 plan
    got = plan.apply(X)
    rows = torch.from_numpy(np.r_[0:6, 1140:1146, T - 6:T]).cuda()          # first / middle / ragged last time block
    Xr = X[rows].cpu().numpy()
    # a 16-region window in every one of the 96 column tiles (= every region block of the entry-list form / every
    # column tile of the tile-sparse form), over ALL 1,036,800 cells
    cols = _tile_windows(R, 16, seed=21)
    ref = c_oracle.dense_synth_cols(Xr, G, R, cols, seed, fill=fill, blocklocal=not uniform)
    _rel_ok(got[rows][:, torch.from_numpy(cols).cuda()].cpu().numpy(), ref, 1e-9 if f64 else RTOL32)   # (fp64 target: 1e-6)
    # denominators against the hashes
    r = 777
    idx = np.arange(G, dtype=np.uint64) * np.uint64(R) + np.uint64(r)
    keep = O.hash_u01(idx, np.uint32(seed) ^ np.uint32(0x9e3779b9)) < np.float32(fill)
    if not uniform:
        keep &= ((97 * (np.arange(G) // 64)) % ((R + 255) // 256)) == r // 256
    np.testing.assert_allclose(plan.den[r], O.hash_u01(idx, seed)[keep].astype(np.float64).sum(), rtol=1e-12)
    # size-independent properties over all 24,378 regions
    const = plan.apply(torch.full((3, G), 7.25, dtype=X.dtype, device="cuda")).cpu().numpy()
    np.testing.assert_allclose(const, 7.25, rtol=1e-12 if f64 else 2e-5)    # constant field -> constant
    assert torch.equal(plan.apply(X * 2.0), got * 2.0)                      # exact linearity in 2x
    short = plan.apply(X[:100].contiguous())                                # rows are independent of the block they sit in
    np.testing.assert_allclose(short.cpu().numpy(), got[:100].cpu().numpy(), rtol=1e-12 if f64 else 2e-5)   # (k is sliced differently)
    plan.close()


# ---------------------------------------------------------------------------------------------
# transforms inside the MFMA forms' pack stage, and the +-inf note
# ---------------------------------------------------------------------------------------------
def test_mfma_forms_pack_stage_transforms_and_inf_note(torch_cuda):
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(12)
    T, G, R = 100, 64 * 30 + 7, 300
    Wf = rng.uniform(0, 1, (G, R)).astype(np.float32)
    Wb = O.blocklocal_weights_oracle(G, R, 5)
    X = (285 + 12 * rng.standard_normal((T, G))).astype(np.float32)
    X[9, 4] = np.nan
    Xhi = X + rng.uniform(0, 9, X.shape).astype(np.float32)
    Xd, Hd = torch.from_numpy(X).cuda(), torch.from_numpy(Xhi).cuda()
    for W, plan in ((Wf, DensePlan.from_host(Wf)), (Wb, DensePlan.synth_blocklocal(G, R, 5))):
        _rel_ok(plan.apply_poly(Xd, -273.15, 3).cpu().numpy(), O.agg_dense(O.tas_poly_values(X, 3), W), RTOL32, scale=1.0)
        edd = O.snyder_edd_values(X + np.float32(-273.15), Xhi + np.float32(-273.15), 14.0)
        _rel_ok(plan.apply_edd(Xd, Hd, 14.0, offset=-273.15).cpu().numpy(), O.agg_dense(edd, W), RTOL32, scale=0.05)
        assert not plan.saw_inf()
        Xi = Xd.clone()
        Xi[3, 100] = float("inf")
        plan.apply(Xi)
        assert plan.saw_inf() and not plan.saw_inf()                         # noted once, then cleared
        plan.apply_poly(torch.full_like(Xd, 1.0e13), 0.0, 3)                  # overflows in the transform
        assert plan.saw_inf()


# ---------------------------------------------------------------------------------------------
# kernel-form flags and the device-failure status
# ---------------------------------------------------------------------------------------------
def test_plan_flags_pin_the_kernel_form(torch_cuda):
    """WAGG_PLAN_* (wagg_plan_create flags): the fp32 kernels (loader/consumer kernel, persistent stream kernel,
    chunk-walking kernel) and both chunk shapes agree.  The dense-tile MFMA consumers (WAGG_PLAN_LC_MFMA) left the production
    library in round 4 (the only kernel with a bounded-spin barrier): the flag is refused there and checked against the
    same numbers in the diagnostic build, in a child process."""
    from climate_toolbox_amd import _lib, synth
    from climate_toolbox_amd.engine import SparsePlan
    torch = torch_cuda
    lat, lon, df = synth.realistic_segments(96, 192, R=300, seed=4, string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
    G = len(lat) * len(lon)
    X = torch.from_numpy((280 + np.random.default_rng(1).standard_normal((130, G))).astype(np.float32)).cuda()
    outs = []
    NL = _lib.PLAN_NO_LINES
    for flags in (0, _lib.PLAN_NO_LC, _lib.PLAN_NO_STREAM, NL, NL | _lib.PLAN_NO_LC, NL | _lib.PLAN_NO_STREAM):
        plan = SparsePlan(cell, code, w, G, len(uniq), row_len=len(lon), flags=flags)
        outs.append(plan.apply(X).cpu().numpy())
        plan.status()
    for o in outs[1:]:
        np.testing.assert_allclose(o, outs[0], rtol=3e-6)
    with pytest.raises(_lib.WaggError):
        SparsePlan(cell, code, w, G, len(uniq), flags=64)
    with pytest.raises(_lib.WaggError, match="diagnostic build"):
        SparsePlan(cell, code, w, G, len(uniq), row_len=len(lon), flags=_lib.PLAN_LC_MFMA)
    diag = os.path.join(ROOT, "climate_toolbox_amd", "lib", "libwagg_diag.so")
    if not os.path.exists(diag):
        return
    ref_path = os.path.join(str(os.environ.get("TMPDIR", "/tmp")), "wagg_flags_ref_%d.npy" % os.getpid())
    np.save(ref_path, outs[0])
    code_ = r'''
import numpy as np, torch
from climate_toolbox_amd import _lib, synth
_lib.LIB_PATH = %r
from climate_toolbox_amd.engine import SparsePlan
lat, lon, df = synth.realistic_segments(96, 192, R=300, seed=4, string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G = len(lat) * len(lon)
X = torch.from_numpy((280 + np.random.default_rng(1).standard_normal((130, G))).astype(np.float32)).cuda()
ref = np.load(%r)
for flags in (_lib.PLAN_LC_MFMA, _lib.PLAN_NO_LINES | _lib.PLAN_LC_MFMA):
    plan = SparsePlan(cell, code, w, G, len(uniq), row_len=len(lon), flags=flags)
    np.testing.assert_allclose(plan.apply(X).cpu().numpy(), ref, rtol=3e-6)
    plan.status()
print("MFMA-CONSUMERS-OK")
''' % (diag, ref_path)
    try:
        r = subprocess.run([sys.executable, "-c", code_], env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=600)
    finally:
        os.remove(ref_path)
    assert "MFMA-CONSUMERS-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("nlat,nlon", [(96, 192), (61, 100), (40, 36)])
def test_whole_line_plan_against_region_shaped_chunks_and_the_oracle(torch_cuda, nlat, nlon):
    """the whole-line chunking (chunks of eight 32-cell lines of one column strip, (chunk, region)
    partial rows, combine kernel; 16-cell lines for fp64) is what plain (time, gridcell) applies of a compact table on a
    grid of known row length use; every other apply, and every apply under WAGG_PLAN_NO_LINES, uses the region-shaped chunks.  Both against the
    fp64 oracle and each other: fp32 and fp64, (time, gridcell) and (gridcell,
    time) data, both result layouts, ragged T, fused powers and degree days, NaN data, rows whose length is not a
    whole number of lines, a split cell, a null label, a region without rows."""
    from climate_toolbox_amd import _lib, synth
    from climate_toolbox_amd.engine import SparsePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    lat, lon, df = synth.realistic_segments(nlat, nlon, R=max(20, nlat * nlon // 70), seed=4, string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "popwt", "hierid")
    cell, code, w = cell.copy(), code.copy(), w.copy()
    code[5] = -1                                                # a null label (S3)
    R = len(uniq) + 1                                           # ... and a last region nobody maps to
    G = nlat * nlon
    rng = np.random.default_rng(2)
    for dtype, rtol in ((np.float32, RTOL32), (np.float64, RTOL64)):
        X = (280 + 15 * rng.standard_normal((135, G))).astype(dtype)
        X[7, cell[11]] = np.nan
        Xi = X.copy()
        Xi[9, cell[20]], Xi[70, cell[31]] = np.inf, -np.inf         # +-inf data: the consumers' general form (S6)
        refi = O.agg_coded(Xi, cell, code, w, R)
        ref = O.agg_coded(X, cell, code, w, R)
        Xd = torch.from_numpy(X).cuda()
        XT = torch.from_numpy(np.ascontiguousarray(X[:50].T)).cuda()
        got = {}
        for flags in (0, _lib.PLAN_NO_LINES):
            plan = SparsePlan(cell, code, w, G, R, row_len=nlon, flags=flags)
            assert plan.info["lines"] == (7 if flags == 0 else 0)       # the fp32, the fp64 and the fp64 degree-day chunking, or none
            if flags == 0:
                assert plan.info["n_partial_rows"] >= len(uniq) - 1 and plan.info["lines_chunks"] > 0
                assert plan.info["n_partial_rows64"] >= plan.info["n_partial_rows"] and plan.info["lines64_chunks"] >= plan.info["lines_chunks"]
                assert plan.info["lines64_ucells"] <= plan.info["lines_ucells"]    # 16-cell lines carry fewer ocean cells along
                assert plan.info["lines_lines128"] * 8 == plan.info["lines_ucells"] // 4 or nlon % 32    # every quad of a line, each line once
            g = plan.apply(Xd).cpu().numpy()
            _rel_ok(g, ref, rtol)
            _rel_ok(plan.apply(torch.from_numpy(Xi).cuda()).cpu().numpy(), refi, rtol)
            # rows that are not 16-byte aligned (a view into a wider buffer): the element-wise loads, same arithmetic
            wide = torch.zeros((X.shape[0], G + 3), dtype=Xd.dtype, device="cuda")
            wide[:, 1:G + 1] = Xd
            np.testing.assert_array_equal(plan.apply(wide[:, 1:G + 1]).cpu().numpy(), g)
            np.testing.assert_array_equal(plan.apply_poly(wide[:, 1:G + 1], -273.15, 3).cpu().numpy(),
                                          plan.apply_poly(Xd, -273.15, 3).cpu().numpy())
            np.testing.assert_array_equal(plan.apply(Xd, out_layout="RT").cpu().numpy(), g.T)
            gt = plan.apply(XT, layout="GT", out_layout="RT").cpu().numpy()
            _rel_ok(gt, ref[:50].T, rtol)
            np.testing.assert_allclose(plan.apply(XT, layout="GT", out_layout="TR").cpu().numpy(), gt.T, rtol=1e-12 if dtype == np.float64 else 1e-6)
            pol = plan.apply_poly(Xd, -273.15, 3).cpu().numpy()
            _rel_ok(pol[1], O.agg_coded(O.tas_poly_values(X, 2), cell, code, w, R), rtol)
            Xhi = X + dtype(5.0)
            edd = O.snyder_edd_values(X + dtype(-273.15), Xhi + dtype(-273.15), 9.0)
            e2 = plan.apply_edd(Xd, torch.from_numpy(Xhi).cuda(), [9.0, 11.0], offset=-273.15).cpu().numpy()
            _rel_ok(e2[0], O.agg_coded(edd, cell, code, w, R), rtol, scale=0.05)
            plan.status()
            got[flags] = g
        np.testing.assert_allclose(got[0], got[_lib.PLAN_NO_LINES], rtol=1e-5 if dtype == np.float32 else 1e-12, equal_nan=True)
        assert np.isnan(got[0][:, R - 1]).all()                 # 0 / 0 for the region without rows (S7)


def test_consumer_barrier_timeout_surfaces_as_an_error(torch_cuda):
    """a consumer-wave barrier that times out inside sparse_lc_kernel must turn
    into WAGG_EHIP, never into silent garbage.  Forced in the diagnostic build only (libwagg_diag.so,
    WAGG_LC_KNOB bit 6: consumer wave 3 stops arriving), in a child process."""
    diag = os.path.join(ROOT, "climate_toolbox_amd", "lib", "libwagg_diag.so")
    if not os.path.exists(diag):
        pytest.skip("diagnostic library not built (make -C climate_toolbox_amd/csrc diag)")
    code = r'''
import numpy as np, torch
from climate_toolbox_amd import _lib, synth
_lib.LIB_PATH = %r
from climate_toolbox_amd.engine import SparsePlan
lat, lon, df = synth.realistic_segments(96, 192, R=300, seed=4, string_labels=False)
cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
G = len(lat) * len(lon)
plan = SparsePlan(cell, code, w, G, len(uniq), row_len=len(lon), flags=_lib.PLAN_LC_MFMA)   # the consumers that have a barrier
X = torch.ones((1920, G), dtype=torch.float32, device="cuda")
plan.apply(X)
try:
    plan.status()
except _lib.WaggError as e:
    print("STATUS-ERROR", e)
try:
    plan.apply(X)
except _lib.WaggError as e:
    print("APPLY-ERROR", e)
''' % diag
    env = dict(os.environ, WAGG_LC_KNOB="64", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "STATUS-ERROR" in r.stdout and "APPLY-ERROR" in r.stdout and "timed out" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


# ---------------------------------------------------------------------------------------------
# fp64 dense-family forms: v_mfma_f64_16x16x4_f64 (the reference's own arithmetic type, S8)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,G,R", [(5, 64, 7), (176, 1000, 130), (177, 4099, 257), (400, 777, 129), (3, 20, 600),
                                   (31, 500, 300), (365, 333, 20), (17, 96, 257)])
def test_dense_f64_small_vs_oracle(torch_cuda, T, G, R):
    """Full fp64 matrix on the f64 MFMA kernel: ragged T (one / several 176-row blocks), G not a
    multiple of the 16-cell tile, R not a multiple of 256; 1e-6 relative (BASELINE), measured ~1e-14."""
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(T * 1000 + G)
    W = rng.uniform(0, 1, (G, R))
    X = 280 + 20 * rng.standard_normal((T, G))
    X[0, 0] = np.nan
    if T > 2:
        X[2, G // 2] = np.nan
    plan = DensePlan.from_host(W)
    assert plan.dtype == "float64" and plan.info["elem_bytes"] == 8 and plan.info["n_kt"] == (G + 15) // 16
    np.testing.assert_allclose(plan.den, W.sum(0), rtol=1e-12)
    ref = O.agg_dense(X, W)
    Xd = torch.from_numpy(X).cuda()
    for ks in (0, 8, 16):
        _rel_ok(plan.apply(Xd, ksplit=ks).cpu().numpy(), ref, 1e-11)
    with pytest.raises(TypeError):
        plan.apply(Xd.float())
    # transforms in the pack stage, in fp64
    _rel_ok(plan.apply_poly(Xd, -273.15, 3).cpu().numpy(), O.agg_dense(O.tas_poly_values(X, 3), W), 1e-10, scale=1.0)
    Xhi = X + rng.uniform(0, 9, X.shape)
    edd = O.snyder_edd_values(X - 273.15, Xhi - 273.15, 14.0)
    _rel_ok(plan.apply_edd(Xd, torch.from_numpy(Xhi).cuda(), 14.0, offset=-273.15).cpu().numpy(), O.agg_dense(edd, W),
            1e-9, scale=0.05)


def test_dense_f64_tile_sparse_and_segments(torch_cuda):
    """Tile-sparse fp64 form: block-local synthetic weights generated on the device, and the same
    table handed over as segments (16-cell x 256-region tiles); against the fp64 oracle."""
    from climate_toolbox_amd import _lib
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    G, R, T, seed = 64 * 75 + 40, 1000, 400, 5
    W = O.blocklocal_weights_oracle(G, R, seed).astype(np.float64)
    plan = DensePlan.synth_blocklocal(G, R, seed, dtype="float64")
    n_kt = (G + 15) // 16
    assert plan.dtype == "float64" and plan.info["form"] == _lib.FORM_TILES and plan.info["n_tiles"] == n_kt
    np.testing.assert_allclose(plan.den, W.sum(0), rtol=1e-12)
    rng = np.random.default_rng(2)
    X = 280 + 20 * rng.standard_normal((T, G))
    X[7, 100] = np.nan
    ref = O.agg_dense(X, W)
    Xd = torch.from_numpy(X).cuda()
    _rel_ok(plan.apply(Xd).cpu().numpy(), ref, 1e-11)
    gi, ri = np.nonzero(W)
    seg = DensePlan.from_segments(gi.astype(np.int32), ri.astype(np.int32), W[gi, ri], G, R, dtype="float64")
    assert seg.dtype == "float64" and seg.info["form"] == _lib.FORM_TILES
    _rel_ok(seg.apply(Xd).cpu().numpy(), ref, 1e-11)
    # regions without any weight: 0/0
    gi2, ri2 = gi[ri < 300], ri[ri < 300]
    part = DensePlan.from_segments(gi2.astype(np.int32), ri2.astype(np.int32), W[gi2, ri2], G, R, dtype="float64")
    out = part.apply(Xd).cpu().numpy()
    assert np.isnan(out[:, 300:]).all()
    _rel_ok(out[:, :300], ref[:, :300], 1e-11)


def test_dropin_takes_the_f64_mfma_form_for_scattered_fp64_tables(torch_cuda):
    """VERDICT r1 missing #1: fp64 data with scattered weights no longer falls to the chunk-walking
    kernel -- the drop-in builds an fp64 dense-family plan (full or tile-sparse) by itself."""
    from climate_toolbox_amd import aggregations as A, minixr
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    rng = np.random.default_rng(3)
    nlat, nlon, R, T = 48, 96, 40, 30
    lat, lon = np.arange(nlat) * 1.0, np.arange(nlon) * 1.0
    n = int(0.3 * nlat * nlon * R)
    flat = rng.choice(nlat * nlon * R, size=n, replace=False)
    cell, lab = flat // R, flat % R
    df = pd.DataFrame({"lat": lat[cell // nlon], "lon": lon[cell % nlon], "areawt": rng.uniform(0.1, 1, n),
                       "popwt": rng.lognormal(0, 1, n), "hierid": lab})
    df.loc[rng.random(n) < 0.2, "popwt"] = np.nan
    tas = 280 + 10 * rng.standard_normal((T, nlat, nlon))
    ds = minixr.Dataset({"tas": (("time", "lat", "lon"), tas)}, coords={"lat": lat, "lon": lon})
    A._PLAN_CACHE.clear()
    out = A.weighted_aggregate_grid_to_regions(ds, "tas", "popwt", "hierid", df)
    assert any(isinstance(p, DensePlan) and p.dtype == "float64" for p in A._PLAN_CACHE.values())
    ref, dims, labs = O.agg_scatter(tas, ("time", "lat", "lon"), lat, lon, df["lat"].values, df["lon"].values,
                                    df["popwt"].values, df["areawt"].values, df["hierid"].values, group_dim="hierid")
    assert out.tas.dims == dims and list(out["hierid"].values) == list(labs)
    _rel_ok(out.tas.values, ref, RTOL64)


def test_dropin_takes_the_f64_entry_list_form_for_sparse_scattered_fp64_tables(torch_cuda):
    """VERDICT r2 missing #1: a c5-uniform-like fp64 table (2 % of the pairs, at random positions, so every tile is
    occupied) no longer needs the full fp64 matrix to fit: the drop-in hands it to the fp64 entry-list form,
    host-resident field (row-block pipeline) and device-resident field alike."""
    from climate_toolbox_amd import _lib, aggregations as A, minixr
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(13)
    nlat, nlon, R, T = 64, 128, 300, 150
    lat, lon = np.arange(nlat) * 1.0, np.arange(nlon) * 1.0
    n = int(0.02 * nlat * nlon * R)
    flat = rng.choice(nlat * nlon * R, size=n, replace=False)
    cell, lab = flat // R, flat % R
    df = pd.DataFrame({"lat": lat[cell // nlon], "lon": lon[cell % nlon], "areawt": rng.uniform(0.1, 1, n),
                       "popwt": rng.lognormal(0, 1, n), "hierid": lab})
    df.loc[rng.random(n) < 0.2, "popwt"] = np.nan
    tas = 280 + 10 * rng.standard_normal((T, nlat, nlon))
    ref, dims, labs = O.agg_scatter(tas, ("time", "lat", "lon"), lat, lon, df["lat"].values, df["lon"].values,
                                    df["popwt"].values, df["areawt"].values, df["hierid"].values, group_dim="hierid")
    for values in (tas, torch.from_numpy(tas).cuda()):
        ds = minixr.Dataset({"tas": (("time", "lat", "lon"), values)}, coords={"lat": lat, "lon": lon})
        A._PLAN_CACHE.clear()
        out = A.weighted_aggregate_grid_to_regions(ds, "tas", "popwt", "hierid", df)
        assert [(type(p), p.dtype, p.info["form"]) for p in A._PLAN_CACHE.values()] == [(DensePlan, "float64", _lib.FORM_ENTRIES)]
        assert out.tas.dims == dims and list(out["hierid"].values) == list(labs)
        _rel_ok(out.tas.values, ref, RTOL64)
        _rel_ok(out.tas.values, ref, 1e-11)


def test_c5_like_fp64_tile_sparse_at_scale(torch_cuda):
    """A c5-like fp64 case at the full grid: block-local weights over G = 1,036,800 cells and
    R = 24,378 regions in the fp64 tile-sparse MFMA form (W 2.1 GB), 365 rows; column windows
    against the C oracle, constant field, exact 2x linearity."""
    from climate_toolbox_amd import engine
    from oracle import c_oracle
    torch = torch_cuda
    G, R, seed, T = 720 * 1440, 24378, 2, 365
    plan = engine.DensePlan.synth_blocklocal(G, R, seed, dtype="float64")
    X = engine.synth_field(T, G, seed=77, base=280.0, amp=60.0, dtype="float64")
    got = plan.apply(X)
    rows = torch.from_numpy(np.r_[0:4, 180:184, T - 4:T]).cuda()
    Xr = X[rows].cpu().numpy()
    for r0 in (0, 11111, R - 16):
        # the C oracle takes fp32 fields: compare on the fp32-rounded rows (its own accumulation is fp64)
        ref = c_oracle.dense_synth_sparse(Xr.astype(np.float32), 0, G, R, r0, 16, seed, fill=0.952, blocklocal=True)
        got32 = plan.apply(torch.from_numpy(Xr.astype(np.float32).astype(np.float64)).cuda()).cpu().numpy()
        _rel_ok(got32[:, r0:r0 + 16], ref, 1e-9)
    const = plan.apply(torch.full((3, G), 7.25, dtype=torch.float64, device="cuda")).cpu().numpy()
    np.testing.assert_allclose(const, 7.25, rtol=1e-13)
    assert torch.equal(plan.apply(X * 2.0), got * 2.0)
    plan.close()


# ---------------------------------------------------------------------------------------------
# host-resident data: row-block pipeline of the C-ABI
# ---------------------------------------------------------------------------------------------
def test_host_streaming_forms_agree_with_the_device_apply(torch_cuda):
    """wagg_apply_host_ex_* / wagg_dense_apply_host_*: whole-field copy, pageable row blocks and
    page-locked row blocks give the bits of the device-resident apply (several blocks, ragged last
    block, padded rows, both dtypes); (gridcell, time) data takes the whole-copy path."""
    from climate_toolbox_amd import _lib, synth
    from climate_toolbox_amd.engine import DensePlan, SparsePlan
    torch = torch_cuda
    lat, lon, df = synth.realistic_segments(360, 720, R=3000, seed=9, string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
    G, R = len(lat) * len(lon), len(uniq)
    plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
    rng = np.random.default_rng(4)
    T = 700                                                    # 259,200 cells x 4 B: 256 MiB ~ 258 rows -> 3 blocks
    for dtype in (np.float32, np.float64):
        X = (280 + 20 * rng.standard_normal((T, G))).astype(dtype)
        X[3, 1000] = np.nan
        ref = plan.apply(torch.from_numpy(X).cuda()).cpu().numpy()
        for flags in (_lib.HOST_WHOLE, 0, _lib.HOST_PIN):
            np.testing.assert_array_equal(plan.apply_host(X, flags=flags), ref)
        XT = np.ascontiguousarray(X[:40].T)                    # (gridcell, time): whole-copy path
        np.testing.assert_array_equal(plan.apply_host(XT, layout="GT", out_layout="RT", flags=_lib.HOST_PIN),
                                      plan.apply(torch.from_numpy(XT).cuda(), layout="GT", out_layout="RT").cpu().numpy())
    with pytest.raises(_lib.WaggError):
        plan.apply_host(X, flags=64)
    # dense-family plan: blocks are whole 368-row launches
    Gd, Rd, Td = 4096, 300, 2100                               # X: 34 MB (page-locked under HOST_PIN), out: 2.5 MB (staged)
    W = rng.uniform(0, 1, (Gd, Rd)).astype(np.float32)
    Xd = (280 + 20 * rng.standard_normal((Td, Gd))).astype(np.float32)
    dplan = DensePlan.from_host(W)
    refd = dplan.apply(torch.from_numpy(Xd).cuda()).cpu().numpy()
    for flags in (_lib.HOST_WHOLE, 0, _lib.HOST_PIN):
        np.testing.assert_allclose(dplan.apply_host(Xd, flags=flags), refd, rtol=2e-6)


def test_host_path_checks_every_status_and_spreads_over_devices(torch_cuda):
    """VERDICT r2 items 1 and 9.  (a) The host path keeps every HIP status: after page-locked, staged and whole-field
    calls the library's counters show every page-lock released again, no refused registration, no failure while
    resources were released -- and which bytes went which way (big arrays in place, small ones through the staging
    pieces; never a runtime copy of a pageable pointer).  (b) wagg_apply_host_multi_* / wagg_dense_apply_host_multi_*:
    the row blocks dealt over several plan replicas -- here two and three pipelines on the one GPU of the box, each
    with its own host thread and streams -- give the bits of the single-device call; misuse is refused."""
    from climate_toolbox_amd import _lib, synth
    from climate_toolbox_amd.engine import DensePlan, SparsePlan
    torch = torch_cuda
    lat, lon, df = synth.realistic_segments(360, 720, R=3000, seed=9, string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
    G, R = len(lat) * len(lon), len(uniq)
    plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
    rng = np.random.default_rng(14)
    T = 1100                                                   # 1.14 GB of fp32 X, a 13 MB result (below the page-lock size)
    X = (280 + 20 * rng.standard_normal((T, G))).astype(np.float32)
    ref = plan.apply(torch.from_numpy(X).cuda()).cpu().numpy()
    _lib.host_stats(reset=True)
    np.testing.assert_array_equal(plan.apply_host(X, flags=_lib.HOST_PIN), ref)
    st = _lib.host_stats()
    assert st["calls"] == 1 and st["blocks"] == _lib.host_block_plan(T, G * 4, 64)[1] == 5
    assert st["registered"] == st["unregistered"] == 1                      # X page-locked in place, the small result staged
    assert st["register_failed"] == st["unregister_failed"] == st["cleanup_failed"] == 0
    assert st["direct_h2d_bytes"] == X.nbytes and st["staged_h2d_bytes"] == 0
    assert st["staged_d2h_bytes"] == ref.nbytes and st["direct_d2h_bytes"] == 0
    np.testing.assert_array_equal(plan.apply_host(X, flags=0), ref)         # no page-lock: everything staged
    st2 = _lib.host_stats()
    assert st2["registered"] == 1 and st2["staged_h2d_bytes"] == X.nbytes and st2["direct_h2d_bytes"] == X.nbytes
    np.testing.assert_array_equal(plan.apply_host(X, flags=_lib.HOST_PIN | _lib.HOST_WHOLE), ref)
    st3 = _lib.host_stats()
    assert st3["registered"] == st3["unregistered"] == 2 and st3["cleanup_failed"] == st3["unregister_failed"] == 0
    # (b) several pipelines: replicas on the same GPU
    for n in (2, 3):
        reps = [plan.replica(0) for _ in range(n - 1)]
        b, nb = _lib.host_block_plan(T, G * 4, 64, n)
        assert nb >= 2 * n
        _lib.host_stats(reset=True)
        for flags in (_lib.HOST_PIN, 0):
            np.testing.assert_array_equal(plan.apply_host(X, flags=flags, replicas=reps), ref)
        st = _lib.host_stats()
        assert st["blocks"] == 2 * nb and st["register_failed"] == st["unregister_failed"] == st["cleanup_failed"] == 0
        assert st["registered"] == st["unregistered"] == 1
        for r in reps:
            r.close()
    X64 = X[:300].astype(np.float64)
    rep = plan.replica(0)
    np.testing.assert_array_equal(plan.apply_host(X64, flags=_lib.HOST_PIN, replicas=[rep]),
                                  plan.apply(torch.from_numpy(X64).cuda()).cpu().numpy())
    with pytest.raises(ValueError):
        plan.apply_host(np.ascontiguousarray(X[:8].T), layout="GT", out_layout="RT", replicas=[rep])
    other = SparsePlan(cell[:100], code[:100] % 7, w[:100], G, 7)
    with pytest.raises(_lib.WaggError, match="another shape"):
        plan.apply_host(X[:64], replicas=[other])
    rep.close()
    # dense-family plans own their workspaces: one replica per pipeline, never the same plan twice
    Gd, Rd, Td = 4096, 300, 2500
    W = rng.uniform(0, 1, (Gd, Rd)).astype(np.float32)
    Xd = (280 + 20 * rng.standard_normal((Td, Gd))).astype(np.float32)
    dplan = DensePlan.from_host(W)
    refd = dplan.apply_host(Xd, flags=_lib.HOST_PIN)
    drep = dplan.replica(0)
    # (another block size means another k split inside the dense-family kernels: same numbers to rounding)
    np.testing.assert_allclose(dplan.apply_host(Xd, flags=_lib.HOST_PIN, replicas=[drep]), refd, rtol=2e-6)
    with pytest.raises(_lib.WaggError, match="same plan"):
        dplan.apply_host(Xd, replicas=[dplan])
    sp = DensePlan.synth(Gd, Rd, 3, fill=0.03, dtype="float64")             # entry lists, fp64
    Xs = Xd.astype(np.float64)
    np.testing.assert_allclose(sp.apply_host(Xs, flags=0, replicas=[sp.replica(0)]), sp.apply_host(Xs, flags=0), rtol=1e-12)


def test_dropin_spreads_host_fields_over_the_listed_devices(torch_cuda):
    """The reference-named function with HOST_DEVICES = [0, 0]: two row-block pipelines (plan + cached replica) serve a
    host-resident field and give the same numbers as one; a second call reuses the replica."""
    from climate_toolbox_amd import _lib, aggregations as A, minixr, synth
    from oracle import ref_numpy as O
    lat, lon, df = synth.realistic_segments(360, 720, R=900, seed=5, string_labels=True)
    rng = np.random.default_rng(15)
    T = 1100
    tas = (280 + 10 * rng.standard_normal((T, len(lat), len(lon)))).astype(np.float32)
    ds = minixr.Dataset({"tas": (("time", "lat", "lon"), tas)}, coords={"time": np.arange(T), "lat": lat, "lon": lon})
    A._PLAN_CACHE.clear()
    one = A.weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df).tas.values
    saved = A.HOST_DEVICES
    try:
        A.HOST_DEVICES = [0, 0]
        _lib.host_stats(reset=True)
        two = A.weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df).tas.values
        (plan,) = A._PLAN_CACHE.values()
        assert len(plan._replicas) == 1 and _lib.host_stats()["blocks"] == _lib.host_block_plan(T, tas[0].nbytes, 64, 2)[1]
        rep = list(plan._replicas.values())[0]
        again = A.weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df).tas.values
        assert list(plan._replicas.values())[0] is rep
    finally:
        A.HOST_DEVICES = saved
    np.testing.assert_array_equal(two, one)
    np.testing.assert_array_equal(again, one)
    st = _lib.host_stats()
    assert st["register_failed"] == st["unregister_failed"] == st["cleanup_failed"] == 0


def test_library_owns_the_data_movement_of_device_fields(torch_cuda):
    """what used to be torch kernels on product branches is the library's own code --
    wagg_combine_planes_* (gdd = EDD(lo) - EDD(hi) of aggregated planes), wagg_take_axis (leap-day drop, lon re-ordering
    of device fields), wagg_relayout_* (the transpose of a field whose lat/lon axes are not adjacent) -- each against
    NumPy, then the drop-in on device-resident (lat, time, lon) fields with a leap day and a 0..360 lon axis against the
    same call on host arrays."""
    from climate_toolbox_amd import engine, minixr, standardize_climate_data, snyder_gdd, weighted_aggregate_grid_to_regions, synth
    from climate_toolbox_amd.transformations import remove_leap_days
    torch = torch_cuda
    rng = np.random.default_rng(31)
    for dt in (np.float32, np.float64):
        a = rng.standard_normal((3, 40, 7, 12)).astype(dt)
        ad = torch.from_numpy(a).cuda()
        np.testing.assert_array_equal(engine.combine_planes(ad, [1.0, -1.0, 0.5]).cpu().numpy(), (a[0] - a[1]) + dt(0.5) * a[2])
        for axis, idx in ((1, np.r_[0:10, 11:40]), (3, rng.permutation(12)), (0, [2, 0])):
            np.testing.assert_array_equal(engine.take_axis(ad, axis, idx).cpu().numpy(), np.take(a, idx, axis=axis))
        for order in ((1, 0, 2, 3), (3, 1, 0, 2), (0, 1, 2, 3)):
            np.testing.assert_array_equal(engine.relayout(ad, order).cpu().numpy(), np.ascontiguousarray(np.transpose(a, order)))
        np.testing.assert_array_equal(engine.relayout(ad[:, ::2, :, 1:9]).cpu().numpy(), a[:, ::2, :, 1:9])
    with pytest.raises(TypeError):
        engine.relayout(torch.zeros((4, 4), dtype=torch.int32, device="cuda"))
    # the drop-in on device-resident fields: (lat, time, lon) order, 0..360 longitudes, a 29 February in the time axis
    nlat, nlon = 48, 96
    lat = np.arange(nlat) * 1.0 - 23.5
    lon360 = np.arange(nlon) * 3.75
    time = pd.date_range("2004-02-20", periods=20).values
    lat_s, lon_s, df = synth.realistic_segments(nlat, nlon, R=60, seed=3, string_labels=False,
                                                lat=lat, lon=np.sort((lon360 + 180) % 360 - 180))
    tmin = (280 + 8 * rng.standard_normal((nlat, len(time), nlon))).astype(np.float32)
    tmax = tmin + rng.uniform(0, 9, tmin.shape).astype(np.float32)
    outs = []
    for wrap in (lambda v: v, lambda v: torch.from_numpy(v).cuda()):
        ds = minixr.Dataset({"tasmin": (("lat", "time", "lon"), wrap(tmin)), "tasmax": (("lat", "time", "lon"), wrap(tmax))},
                            coords={"lat": lat, "time": time, "lon": lon360})
        ds = remove_leap_days(standardize_climate_data(ds))
        ds["gdd"] = snyder_gdd(ds["tasmin"], ds["tasmax"], 283.15, 303.15)
        outs.append(weighted_aggregate_grid_to_regions(ds, "gdd", "areawt", "hierid", df).gdd.values)
    assert outs[0].shape[1] == 19                                              # (region, time) with the leap day gone
    np.testing.assert_allclose(outs[1], outs[0], rtol=2e-5, atol=1e-6)


def test_dropin_from_two_threads_shares_the_caches_safely(torch_cuda):
    """the module-level caches are locked and a plan handed out is leased.  Two Python threads call
    the reference-named function at once -- the same table (they serialise on the plan's lease), two different tables
    (they run concurrently), with a cache of ONE plan so that every miss tries to evict the other thread's plan while it
    is being applied -- and every result equals the single-threaded one."""
    import threading
    from climate_toolbox_amd import aggregations as A, minixr, synth
    lat, lon, dfa = synth.realistic_segments(96, 192, R=300, seed=4, string_labels=True)
    _, _, dfb = synth.realistic_segments(96, 192, R=200, seed=9, string_labels=True)
    rng = np.random.default_rng(21)
    T = 90
    fields = [(280 + 10 * rng.standard_normal((T, len(lat), len(lon)))).astype(dt) for dt in (np.float32, np.float64)]
    jobs = [(f, df, wt) for f in fields for df, wt in ((dfa, "areawt"), (dfb, "popwt"), (dfa, "popwt"))]

    def run(job):
        f, df, wt = job
        ds = minixr.Dataset({"tas": (("time", "lat", "lon"), f)}, coords={"time": np.arange(T), "lat": lat, "lon": lon})
        return A.weighted_aggregate_grid_to_regions(ds, "tas", wt, "hierid", df).tas.values

    A._PLAN_CACHE.clear()
    expect = [run(j) for j in jobs]
    saved = A._plans._PLAN_CACHE_MAX
    errs, got = [], {}

    def worker(k):
        try:
            for rep in range(6):
                for i in (range(len(jobs)) if k == 0 else reversed(range(len(jobs)))):
                    got[(k, rep, i)] = run(jobs[i])
        except Exception as e:                                            # pragma: no cover
            errs.append(repr(e))

    try:
        A._plans._PLAN_CACHE_MAX = 1
        A._PLAN_CACHE.clear()
        ts = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    finally:
        A._plans._PLAN_CACHE_MAX = saved
    assert not errs, errs
    assert len(got) == 2 * 6 * len(jobs)
    for (k, rep, i), v in got.items():
        np.testing.assert_array_equal(v, expect[i])
    assert len(A._PLAN_CACHE) <= 2                                        # (a leased plan may outlive the limit by one)


def test_dropin_streams_host_fields_through_the_pipeline(torch_cuda):
    """A plain aggregation of a NumPy-backed (time, lat, lon) variable goes through the row-block
    pipeline (no whole-field device copy) and matches the oracle; a device-resident variable and a lazily
    transformed one take the device path and give the same numbers."""
    from climate_toolbox_amd import minixr, weighted_aggregate_grid_to_regions
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(6)
    lat, lon = np.arange(-44.5, 45, 1.0), np.arange(-89.5, 90, 1.0)
    T, n = 130, 4000
    tas = (280 + 10 * rng.standard_normal((T, len(lat), len(lon)))).astype(np.float32)
    df = pd.DataFrame({"lat": rng.choice(lat, n), "lon": rng.choice(lon, n), "areawt": rng.uniform(0.1, 1, n),
                       "hierid": rng.integers(0, 60, n)})
    ref = O.agg_scatter(tas, ("time", "lat", "lon"), lat, lon, df["lat"].values, df["lon"].values, df["areawt"].values,
                        df["areawt"].values, df["hierid"].values, group_dim="hierid")[0]
    coords = {"time": np.arange(T), "lat": lat, "lon": lon}
    host = weighted_aggregate_grid_to_regions(minixr.Dataset({"tas": (("time", "lat", "lon"), tas)}, coords=coords),
                                              "tas", "areawt", "hierid", df).tas.values
    dev = weighted_aggregate_grid_to_regions(
        minixr.Dataset({"tas": minixr.DataArray(torch.from_numpy(tas).cuda(), ("time", "lat", "lon"))}, coords=coords),
        "tas", "areawt", "hierid", df).tas.values
    _rel_ok(host, ref, RTOL32)
    np.testing.assert_array_equal(host, dev)


# ---------------------------------------------------------------------------------------------
# tile-sparse form: pack-free first pass (the MFMA kernel reads X where it lies) + gated exact second pass
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_tile_sparse_pack_free_pass_and_its_exact_fallback(torch_cuda, dtype):
    """Aligned plain applies of a tile-sparse plan skip dense_pack_x_kernel: same bits as the packed pass
    (forced here through a 4-byte-shifted view of the same numbers), against the oracle, for ragged T
    (clamped rows), padded row pitch, and -- through the device-gated second pass -- NaN (S6: skipped)
    and +-inf data (noted for the caller)."""
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    G, R, seed = 64 * 80, 1000, 5                              # whole k tiles: the pack-free pass applies
    W = O.blocklocal_weights_oracle(G, R, seed)
    plan = DensePlan.synth_blocklocal(G, R, seed, dtype="float64" if dtype == np.float64 else "float32")
    assert plan.info["tiled"] == 1
    rng = np.random.default_rng(12)
    rtol = RTOL32 if dtype == np.float32 else 1e-9

    def shifted(Xh):
        """the same rows at an address that is not 16-byte aligned -> the packed pass"""
        big = torch.zeros((Xh.shape[0], G + 4), dtype=tdt, device="cuda")
        view = big[:, 1:G + 1]
        view.copy_(torch.from_numpy(Xh))
        assert view.data_ptr() % 16 != 0
        return view

    for T in (400, 1, 17, 209):
        Xh = (280 + 20 * rng.standard_normal((T, G))).astype(dtype)
        ref = O.agg_dense(Xh, W)
        got = plan.apply(torch.from_numpy(Xh).cuda())
        _rel_ok(got.cpu().numpy(), ref, rtol)
        assert torch.equal(got, plan.apply(shifted(Xh)))       # pack-free == packed, bit for bit
        assert not plan.saw_inf()
        # padded pitch (rows 16-byte aligned, ldx > G)
        pitched = torch.zeros((T, G + 64), dtype=tdt, device="cuda")
        pitched[:, :G].copy_(torch.from_numpy(Xh))
        pitched[:, G:] = float("nan")                          # never read as data of a stored cell
        assert torch.equal(plan.apply(pitched[:, :G]), got)
    # NaN data: the first pass trips the gate, the second pass gives the skipna result
    T = 300
    Xh = (280 + 20 * rng.standard_normal((T, G))).astype(dtype)
    Xh[7, 100] = np.nan
    Xh[T - 1, G - 1] = np.nan
    ref = O.agg_dense(Xh, W)
    got = plan.apply(torch.from_numpy(Xh).cuda())
    _rel_ok(got.cpu().numpy(), ref, rtol)
    assert torch.equal(got, plan.apply(shifted(Xh)))
    assert not plan.saw_inf()
    # the gate is per apply: a finite field afterwards takes the first pass only and is right
    Xf = np.nan_to_num(Xh, nan=281.0)
    _rel_ok(plan.apply(torch.from_numpy(Xf).cuda()).cpu().numpy(), O.agg_dense(Xf, W), rtol)
    # +-inf data: same note for the caller as the packed pass gives
    Xh[5, 64] = np.inf
    plan.apply(torch.from_numpy(Xh).cuda())
    assert plan.saw_inf()
    assert not plan.saw_inf()
    plan.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_full_form_pack_free_pass_for_tall_row_blocks(torch_cuda, dtype):
    """The full-matrix form takes the same pack-free first pass when the row blocks are tall (fp32: >= 20 x 16 rows,
    fp64: >= 10 x 16): same bits as the packed pass, NaN / inf through the gated second pass; short blocks and
    ragged grids keep the packed pass."""
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    rng = np.random.default_rng(21)
    G, R = 4096, 300
    W = rng.uniform(0.0, 1.0, (G, R)).astype(dtype)
    plan = DensePlan.from_host(W)
    assert plan.info["tiled"] == 0
    rtol = RTOL32 if dtype == np.float32 else 1e-9
    tall = (700, 1369, 365) if dtype == np.float32 else (340, 500, 170)     # two or more tall row blocks; one block (365 | 170): packed

    def shifted(Xh):
        big = torch.zeros((Xh.shape[0], G + 4), dtype=tdt, device="cuda")
        view = big[:, 1:G + 1]
        view.copy_(torch.from_numpy(Xh))
        return view

    for T in tall + (40,):
        Xh = (280 + 20 * rng.standard_normal((T, G))).astype(dtype)
        got = plan.apply(torch.from_numpy(Xh).cuda())
        _rel_ok(got.cpu().numpy(), O.agg_dense(Xh, W), rtol)
        assert torch.equal(got, plan.apply(shifted(Xh)))
    T = tall[0]
    Xh = (280 + 20 * rng.standard_normal((T, G))).astype(dtype)
    Xh[3, 17] = np.nan
    Xh[T - 1, G - 1] = np.nan
    got = plan.apply(torch.from_numpy(Xh).cuda())
    _rel_ok(got.cpu().numpy(), O.agg_dense(Xh, W), rtol)
    assert torch.equal(got, plan.apply(shifted(Xh)))
    assert not plan.saw_inf()
    Xh[9, 1000] = -np.inf
    plan.apply(torch.from_numpy(Xh).cuda())
    assert plan.saw_inf()
    plan.close()


# ---------------------------------------------------------------------------------------------
# launch-bound small cases: applies can be captured into a HIP graph and replayed
# ---------------------------------------------------------------------------------------------
def test_applies_are_graph_capturable_after_one_warm_up(torch_cuda):
    """After one apply on a stream (which creates the plan's per-stream staging), the same apply enqueues
    kernels and async memsets only: it can be captured on that stream and replayed on new data (c1-sized
    segment table, fused tas_poly, tile-sparse plan with its gated second pass)."""
    from climate_toolbox_amd import synth
    from climate_toolbox_amd.engine import DensePlan, SparsePlan
    torch = torch_cuda
    lat, lon, tas, df = synth.c1_workload(T=365)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
    G, R, T = len(lat) * len(lon), len(uniq), 365
    plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
    dplan = DensePlan.synth_blocklocal(64 * 80, 700, 3)
    rng = np.random.default_rng(8)
    X = torch.from_numpy(np.ascontiguousarray(tas.reshape(T, G))).cuda()
    X32 = X.float()
    Xd = torch.from_numpy((280 + 10 * rng.standard_normal((300, 64 * 80))).astype(np.float32)).cuda()
    out = torch.empty((T, R), dtype=X.dtype, device="cuda")
    pout = torch.empty((3, T, R), dtype=torch.float32, device="cuda")
    dout = torch.empty((300, 700), dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                                   # warm-up on the capture stream
        plan.apply(X, out=out); plan.apply_poly(X32, -273.15, 3, out=pout); dplan.apply(Xd, out=dout)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        plan.apply(X, out=out); plan.apply_poly(X32, -273.15, 3, out=pout); dplan.apply(Xd, out=dout)
    for scale in (1.0, 0.5, 3.0):
        X.mul_(scale); X32.mul_(scale); Xd.mul_(scale)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        got, gotp, gotd = out.clone(), pout.clone(), dout.clone()
        assert torch.equal(got, plan.apply(X))
        assert torch.equal(gotp, plan.apply_poly(X32, -273.15, 3))
        assert torch.equal(gotd, dplan.apply(Xd))
    Xd[5, 7] = float("nan")                                      # the gated second pass is part of the graph
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(dout, dplan.apply(Xd)) and not torch.isnan(dout).any()
    plan.status()
    dplan.close()


def test_ragged_grid_inside_a_padded_pitch_never_reads_the_pad_cells(torch_cuda):
    """G is not a whole number of k tiles: whatever the row pitch offers behind the G cells (here: room for the whole
    last tile, filled with NaN) is never read -- such grids keep the packed pass."""
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    G, R, T, seed = 64 * 80 + 8, 600, 250, 9
    W = O.blocklocal_weights_oracle(G, R, seed)
    plan = DensePlan.synth_blocklocal(G, R, seed)
    assert plan.info["tiled"] == 1
    rng = np.random.default_rng(3)
    Xh = (280 + 20 * rng.standard_normal((T, G))).astype(np.float32)
    ref = O.agg_dense(Xh, W)
    pitched = torch.zeros((T, G + 24), dtype=torch.float32, device="cuda")      # pitch = 161 k tiles of 32 cells
    pitched[:, :G].copy_(torch.from_numpy(Xh))
    got = plan.apply(pitched[:, :G])
    _rel_ok(got.cpu().numpy(), ref, RTOL32)
    exact = plan.apply(torch.from_numpy(Xh).cuda())             # pitch = G: not whole k tiles -> packed pass
    assert torch.equal(got, exact)
    pitched[:, G:] = float("nan")
    assert torch.equal(plan.apply(pitched[:, :G]), exact)
    assert not plan.saw_inf()
    plan.close()


def _guarded(count, dtype):
    """numpy array of `count` elements whose last byte sits right in front of a PROT_NONE page."""
    import ctypes, mmap
    pg = mmap.PAGESIZE
    nbytes = count * np.dtype(dtype).itemsize
    total = (nbytes + pg - 1) // pg * pg + pg
    m = mmap.mmap(-1, total)
    addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert libc.mprotect(addr + total - pg, pg, 0) == 0                      # PROT_NONE
    return np.frombuffer(m, dtype=dtype, count=count, offset=total - pg - nbytes), m


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_host_forms_respect_the_pitch_of_host_arrays(torch_cuda, dtype):
    """C-ABI contract of the *_host_* entry points for pitched arrays (ldx > G, ldo > R): nothing behind the last
    row's G cells is read (the array ends at an inaccessible page), the padding between result rows is left as it
    was -- whole-copy, staged-block and page-locked-block forms, segment-table and dense plans."""
    import ctypes as C
    from climate_toolbox_amd import _lib, synth
    from climate_toolbox_amd.engine import DensePlan, SparsePlan
    torch = torch_cuda
    L = _lib.load()
    sfx = "f32" if dtype == np.float32 else "f64"
    lat, lon, df = synth.realistic_segments(64, 80, R=300, seed=2, string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
    G, R, T = len(lat) * len(lon), len(uniq), 70
    ldx, ldo = G + 96, R + 20
    rng = np.random.default_rng(1)
    Xc = (280 + 20 * rng.standard_normal((T, G))).astype(dtype)
    X, keep_x = _guarded((T - 1) * ldx + G, dtype)
    X[:] = -1e30
    for t in range(T):
        X[t * ldx:t * ldx + G] = Xc[t]
    splan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
    W = rng.uniform(0, 1, (G, R)).astype(dtype)
    dplan = DensePlan.from_host(W)
    refs = {"sparse": splan.apply(torch.from_numpy(Xc).cuda()).cpu().numpy(),
            "dense": dplan.apply(torch.from_numpy(Xc).cuda()).cpu().numpy()}
    vp = lambda a: C.c_void_p(a.ctypes.data)
    for flags in (_lib.HOST_WHOLE, 0, _lib.HOST_PIN):
        for kind in ("sparse", "dense"):
            out, keep_o = _guarded((T - 1) * ldo + R, dtype)
            out[:] = 777.0
            if kind == "sparse":
                rc = getattr(L, "wagg_apply_host_ex_" + sfx)(splan._h, vp(X), T, ldx, 0, vp(out), ldo, 0, flags)
            else:
                rc = getattr(L, "wagg_dense_apply_host_" + sfx)(dplan._h, vp(X), T, ldx, vp(out), ldo, flags)
            assert rc == 0, L.wagg_last_error()
            rows = np.stack([out[t * ldo:t * ldo + R] for t in range(T)])
            if kind == "sparse":
                np.testing.assert_array_equal(rows, refs[kind])
            else:
                np.testing.assert_allclose(rows, refs[kind], rtol=2e-6 if dtype == np.float32 else 1e-12)
            pads = np.concatenate([out[t * ldo + R:(t + 1) * ldo] for t in range(T - 1)])
            assert (pads == 777.0).all()                       # the caller's padding is untouched
            del out, rows, pads
            keep_o.close()
    del X
    keep_x.close()
    dplan.close()


def test_results_come_back_through_recycled_page_locked_memory(torch_cuda):
    """The drop-in's D2H copy: through a pool of page-locked blocks while the cap allows; a block returns to the pool when
    the caller drops the result (views included) and is handed out again as it is; beyond the cap the pageable copy; same
    numbers either way."""
    import gc
    from climate_toolbox_amd import aggregations as A, minixr, synth, weighted_aggregate_grid_to_regions
    torch = torch_cuda
    lat, lon, df = synth.realistic_segments(360, 720, R=3000, seed=4, string_labels=False)
    T = 120
    X = (280 + 10 * np.random.default_rng(0).standard_normal((T, len(lat), len(lon)))).astype(np.float32)
    ds = minixr.Dataset({"tas": (("time", "lat", "lon"), torch.from_numpy(X).cuda())}, coords={"lat": lat, "lon": lon})
    gc.collect()
    A.clear_caches()                                            # (free blocks of other shapes left by earlier tests go: the
    pool = A._PINNED_POOL                                       #  accounting below is about this test's blocks only)
    n_free = lambda: sum(len(v) for v in pool["free"].values())
    base, free0 = pool["bytes"], n_free()
    a = weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
    n = a.tas.values.nbytes
    size = (n + (1 << 20) - 1) >> 20 << 20
    took_new = pool["bytes"] - base                             # 0 if an earlier test left a free block of this size
    assert n >= 1 << 20 and took_new in (0, size)
    b = weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
    in_use_bytes = pool["bytes"]
    np.testing.assert_array_equal(a.tas.values, b.tas.values)
    pa = a.tas.values.ctypes.data
    keep = a.tas.values[3:5]                                    # a view keeps its block out of the pool
    del a, b
    gc.collect()
    free1 = n_free()                                            # b's block is back, a's is not
    c = weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
    assert pool["bytes"] == in_use_bytes and n_free() == free1 - 1          # a pooled block was reused: nothing new page-locked
    assert c.tas.values.ctypes.data != pa
    np.testing.assert_array_equal(c.tas.values[3:5], keep)
    old_cap = A._pinned._PINNED_OUT_CAP
    try:
        A._pinned._PINNED_OUT_CAP = 0                                   # at the cap with every block in use: pageable copy, same numbers
        for blocks in pool["free"].values():
            blocks.clear()
        d = weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
        np.testing.assert_array_equal(d.tas.values[3:5], keep)
        assert pool["bytes"] == in_use_bytes and n_free() == 0
    finally:
        A._pinned._PINNED_OUT_CAP = old_cap
    del keep, c, d
    gc.collect()
    assert n_free() >= 2                                        # both blocks (a's via its view, c's) came back
    assert free0 >= 0
    A.clear_caches()                                            # plans, memos and the free blocks of the pool go
    assert n_free() == 0 and pool["bytes"] == 0 and len(A._PLAN_CACHE) == 0
    e = weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)      # ... and everything comes back on demand
    np.testing.assert_array_equal(e.tas.values[3:5], weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df).tas.values[3:5])


@pytest.mark.parametrize("layout", ["TG", "GT"])
def test_fp64_degree_days_in_the_loader_consumer_kernel_keep_fp64_accuracy(torch_cuda, layout):
    """fp64 Snyder degree days of finite fields run in sparse_lcv_kernel on the 64-cell chunking with the band value in
    the form w (1 - |z|)^(3/2) P(|z|) + max(d, 0) (degree-13 fit of P, reciprocal and square root by one Newton step:
    csrc/wagg_common.h::snyder_edd1_finite) instead of libm's asin: the result must still be an fp64 result -- 1e-12
    against the oracle's formula (transformations.py:64-87), including cells with tasmin == tasmax and thresholds below,
    inside and above the data."""
    from climate_toolbox_amd.engine import SparsePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(5)
    nlat, nlon, T = 64, 128, 130
    G = nlat * nlon
    cell = np.arange(G, dtype=np.int32)
    code = ((cell // nlon) // 8 * (nlon // 16) + (cell % nlon) // 16).astype(np.int32)
    R = int(code.max()) + 1
    w = rng.uniform(0.1, 1, G)
    tmin = rng.uniform(-20, 40, (T, G)) + 273.15
    tmax = tmin + rng.uniform(0, 20, (T, G))
    tmax[3, :50] = tmin[3, :50]                                  # zero daily range
    plan = SparsePlan(cell, code, w, G, R, row_len=nlon)
    assert plan.info["lines"] == 7                               # the 64-cell chunking is there
    thr = [-30.0, 5.0, 17.3, 31.0, 70.0]                         # (five: a pass of four and one of one)
    a, b = (tmin, tmax) if layout == "TG" else (np.ascontiguousarray(tmin.T), np.ascontiguousarray(tmax.T))
    got = plan.apply_edd(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), thr, offset=-273.15, layout=layout).cpu().numpy()
    for k, e in enumerate(thr):
        ref = O.agg_coded(O.snyder_edd_values(tmin - 273.15, tmax - 273.15, e), cell, code, w, R)
        _rel_ok(got[k], ref, 1e-12, scale=1e-3)
    plan.status()
