"""GPU parity tests added in round 4 (run with -m gpu on an MI355X): weight tables HANDED IN BY THE CALLER -- COO segment
rows and CSR (BASELINE configs[4] "sparse CSR weights") -- built into plans on the device, up to the full configs[4]
size, where the plan must equal the one the on-device hash generator builds, bit for bit.  Every comparison goes through
the C-ABI; the oracle is only the checker (and the independent generator of the caller's arrays)."""
import os
import time

import numpy as np
import pytest

from tests.test_gpu_parity import RTOL32, _rel_ok

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


def _csr_of(cell, code, w, G, rng=None):
    """CSR arrays of a COO table: rows = cells, the entries of a row in table order (so repeated pairs keep their order),
    or shuffled inside the rows when ``rng`` is given (columns of a row may come in any order)."""
    order = np.argsort(cell, kind="stable")
    if rng is not None:
        order = order[np.lexsort((rng.random(len(order)), cell[order]))]
    rowptr = np.zeros(G + 1, dtype=np.int64)
    np.add.at(rowptr, cell + 1, 1)
    return np.cumsum(rowptr), code[order].astype(np.int32), w[order].astype(np.float64)


def _table(kind, G, R, rng):
    """A caller's table with everything S3-S5 allow: repeated (cell, region) rows, null labels, NaN weights, no order."""
    if kind == "scattered":                           # ~2 % of the pairs, every tile touched -> entry lists
        n = int(0.02 * G * R)
        flat = rng.choice(G * R, size=n, replace=False)
        cell, code = flat // R, flat % R
    elif kind == "blocks":                            # each run of 64 cells touches one column tile -> tile-sparse
        n_nt = (R + 255) // 256
        cell = np.repeat(np.arange(G), 12)
        tile = (97 * (cell // 64)) % n_nt
        code = np.minimum(tile * 256 + rng.integers(0, 256, len(cell)), R - 1)
    else:                                             # 30 % of all pairs -> full matrix
        n = int(0.3 * G * R)
        flat = rng.choice(G * R, size=n, replace=False)
        cell, code = flat // R, flat % R
    w = rng.uniform(0.1, 1.0, len(cell))
    dup = rng.integers(0, len(cell), len(cell) // 7)  # repeated pairs (S5): they add, in table order
    cell = np.concatenate([cell, cell[dup]]); code = np.concatenate([code, code[dup]]); w = np.concatenate([w, rng.uniform(0.1, 1.0, len(dup))])
    perm = rng.permutation(len(cell))
    cell, code, w = cell[perm], code[perm].astype(np.int64), w[perm]
    code[rng.random(len(code)) < 0.01] = -1           # null labels leave both sums (S3)
    w[rng.random(len(w)) < 0.01] = np.nan             # NaN weights too (S4)
    return cell.astype(np.int32), code.astype(np.int32), w


@pytest.mark.parametrize("dtype,rtol", [(np.float32, RTOL32), (np.float64, 1e-11)])
@pytest.mark.parametrize("kind,G,R,form", [("scattered", 64 * 96 + 5, 1600, 2), ("scattered", 300, 5, 2), ("blocks", 64 * 40, 700, 1),
                                           ("dense", 1537, 96, 0), ("blocks", 64 * 9 + 3, 600, 1)])
def test_tables_built_on_the_device_vs_oracle(torch_cuda, kind, G, R, form, dtype, rtol):
    """wagg_dense_create_from_segments* / _from_csr* (sort, coalesce, denominators, form choice and packing on the device,
    csrc/wagg_build.hip) against the oracle's aggregation of the same rows (aggregations.py:73-80), all three forms."""
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(G + R)
    cell, code, w = _table(kind, G, R, rng)
    T = 70
    X = (280 + 15 * rng.standard_normal((T, G))).astype(dtype)
    X[6, G // 3] = np.nan
    ref = O.agg_coded(X, cell, code, w, R)
    keep = (code >= 0) & ~np.isnan(w)
    den = np.bincount(code[keep], weights=w[keep], minlength=R)
    Xd = torch.from_numpy(X).cuda()
    # (the form is pinned: these small tables exist to put every packing route through the oracle; what the library picks by
    #  itself -- by estimated time, tests/test_gpu_round5.py -- is built as well and must give the same numbers)
    fname = {0: "full", 1: "tiles", 2: "entries"}[form]
    seg = DensePlan.from_segments(cell, code, w, G, R, dtype=dtype, form=fname)
    assert seg.info["form"] == form and seg.dtype == np.dtype(dtype).name
    assert seg.info["nnz"] == len(np.unique(cell[keep].astype(np.int64) * R + code[keep]))
    np.testing.assert_allclose(seg.den, den, rtol=1e-13)
    got = seg.apply(Xd).cpu().numpy()
    _rel_ok(got, ref, rtol)
    # the same table as CSR (rows in table order inside a cell: repeated pairs add in the same order) -> the same plan, bit for bit
    csr = DensePlan.from_csr(*_csr_of(cell, code, w, G), G, R, dtype=dtype, form=fname)
    assert csr.info["form"] == form and csr.info["nnz"] == seg.info["nnz"]
    np.testing.assert_array_equal(csr.den, seg.den)
    np.testing.assert_array_equal(csr.apply(Xd).cpu().numpy(), got)
    # ... and with the columns of every row in a random order: still the oracle's numbers
    mix = DensePlan.from_csr(*_csr_of(cell, code, w, G, rng), G, R, dtype=dtype, form=fname)
    _rel_ok(mix.apply(Xd).cpu().numpy(), ref, rtol)
    auto = DensePlan.from_segments(cell, code, w, G, R, dtype=dtype)              # the library's own choice of form
    assert auto.info["form"] in (0, 1, 2) and auto.info["nnz"] == seg.info["nnz"] and min(auto.info["est_full_s"], auto.info["est_tiles_s"],
                                                                                       auto.info["est_entries_s"]) > 0
    np.testing.assert_array_equal(auto.den, seg.den)
    _rel_ok(auto.apply(Xd).cpu().numpy(), ref, rtol)
    # a second build of the same table is the same plan (stable sort, fixed summation orders, no atomics on data)
    again = DensePlan.from_segments(cell, code, w, G, R, dtype=dtype, form=fname)
    np.testing.assert_array_equal(again.den, seg.den)
    np.testing.assert_array_equal(again.apply(Xd).cpu().numpy(), got)
    assert seg.info["build_s"] > 0 and 0 <= seg.info["build_upload_s"] <= seg.info["build_s"]


def test_table_builder_edge_cases(torch_cuda):
    """Empty tables, tables that drop every row, one row; bad row offsets and out-of-range codes are rejected with a status."""
    from climate_toolbox_amd import _lib
    from climate_toolbox_amd.engine import DensePlan
    torch = torch_cuda
    G, R = 500, 20
    X = torch.ones((3, G), dtype=torch.float32, device="cuda")
    e = np.zeros(0, np.int32)
    for plan in (DensePlan.from_segments(e, e, np.zeros(0), G, R), DensePlan.from_csr(np.zeros(G + 1, np.int64), e, np.zeros(0), G, R),
                 DensePlan.from_segments(np.array([3, 4], np.int32), np.array([-1, 2], np.int32), np.array([1.0, np.nan]), G, R)):
        assert plan.info["nnz"] == 0 and (plan.den == 0).all()
        assert torch.isnan(plan.apply(X)).all()                                    # 0 / 0 (S7)
    one = DensePlan.from_segments(np.array([7], np.int32), np.array([5], np.int32), np.array([0.25]), G, R, dtype="float64")
    out = one.apply(X.double() * 3.0).cpu().numpy()
    assert (out[:, 5] == 3.0).all() and np.isnan(np.delete(out, 5, axis=1)).all()
    rp = np.zeros(G + 1, np.int64); rp[5:] = 2; rp[9] = 1
    with pytest.raises(_lib.WaggError, match="rowptr decreases at row 8"):
        DensePlan.from_csr(rp, np.zeros(2, np.int32), np.ones(2), G, R)
    with pytest.raises(ValueError):
        DensePlan.from_csr(np.zeros(G + 1, np.int64), np.zeros(2, np.int32), np.ones(2), G, R)
    with pytest.raises(_lib.WaggError, match="segment 1 out of range"):
        DensePlan.from_segments(np.array([1, 2, 3], np.int32), np.array([0, R, 1], np.int32), np.ones(3), G, R)
    with pytest.raises(_lib.WaggError, match="segment 2 out of range"):
        DensePlan.from_segments(np.array([1, 2, G], np.int32), np.array([0, 1, 1], np.int32), np.ones(3), G, R)


@pytest.mark.parametrize("blocklocal,fill", [(False, 0.02), (True, 0.952)])
def test_device_table_generator_equals_the_oracle_generator(torch_cuda, blocklocal, fill):
    """wagg_synth_table_csr (the benchmark's stand-in for a caller's file) against the oracle's generator of the same
    hashes: identical arrays, and -- uniform structure -- against the NumPy restatement of the hash."""
    from climate_toolbox_amd import engine
    from oracle import c_oracle, ref_numpy as O
    G, R, seed = 64 * 123 + 17, 2500, 9
    rp, col, val = engine.synth_table_csr(G, R, seed, fill, blocklocal=blocklocal)
    rp2, col2, val2 = c_oracle.synth_csr(G, R, seed, fill, blocklocal=blocklocal)
    np.testing.assert_array_equal(rp, rp2)
    np.testing.assert_array_equal(col, col2)
    np.testing.assert_array_equal(val, val2)
    W = O.blocklocal_weights_oracle(G, R, seed, fill) if blocklocal else O.dense_weights_oracle(G, R, seed, fill)
    gi, ri = np.nonzero(W)
    np.testing.assert_array_equal(np.repeat(np.arange(G), np.diff(rp)), gi)
    np.testing.assert_array_equal(col, ri)
    np.testing.assert_array_equal(val, W[gi, ri].astype(np.float64))


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[4] with the table handed in by the caller, at full size
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("structure", ["uniform", "block_local"])
def test_c5_table_from_the_caller_at_full_size(torch_cuda, structure):
    """G = 1,036,800 cells x R = 24,378 regions, ~2.5e8 pairs as host CSR arrays (3 GB) from the ORACLE's generator ->
    wagg_dense_create_from_csr[_f64]: the plan has the form, the pair count and the denominators of the plan that
    wagg_dense_create_synth_sparse / _synth_blocklocal generates from the same hashes on the device, and its apply over
    the full rank shard (T = 2,282 of 18,250 rows) is bit-equal -- fp32 and fp64.  (The synthetic plans themselves are
    checked against the oracle's arithmetic in test_c5_full_size_rank_shard.)"""
    from climate_toolbox_amd import _lib, engine
    from climate_toolbox_amd.timeshard import shard_bounds
    from oracle import c_oracle
    torch = torch_cuda
    G, R, seed = 720 * 1440, 24378, 2
    T = shard_bounds(50 * 365, 8)[0][1]
    uniform = structure == "uniform"
    fill = 0.01 if uniform else 0.952
    t0 = time.time()
    rowptr, col, val = c_oracle.synth_csr(G, R, seed, fill, blocklocal=not uniform)
    t_gen = time.time() - t0
    assert abs(len(col) - 0.01 * G * R) < 1e-2 * 0.01 * G * R
    X32 = engine.synth_field(T, G, seed=1000, base=280.0, amp=60.0)
    for dtype in ("float32", "float64"):
        if not uniform and dtype == "float64":
            X = X32[:600].double()                      # (the fp64 tile-sparse plan at 600 rows: the same code, a third of the time)
        else:
            X = X32 if dtype == "float32" else X32.double()
        t0 = time.time()
        plan = engine.DensePlan.from_csr(rowptr, col, val, G, R, dtype=dtype)
        t_build = time.time() - t0
        ref = (engine.DensePlan.synth(G, R, seed, fill=fill, dtype=dtype) if uniform
               else engine.DensePlan.synth_blocklocal(G, R, seed, fill=fill, dtype=dtype))
        assert plan.info["form"] == ref.info["form"] == (_lib.FORM_ENTRIES if uniform else _lib.FORM_TILES)
        assert plan.info["nnz"] == len(col) and plan.info["w_bytes"] == ref.info["w_bytes"] and plan.info["n_tiles"] == ref.info["n_tiles"]
        if uniform:
            assert ref.info["nnz"] == len(col)
        np.testing.assert_array_equal(plan.den, ref.den)        # (24-bit weights: every fp64 summation order is exact)
        got, want = plan.apply(X), ref.apply(X)
        assert torch.equal(got, want)
        print("c5-%s %s: oracle generator %.1f s, from_csr %.2f s (library: %.2f s, of which upload %.2f s)" % (
            structure, dtype, t_gen, t_build, plan.info["build_s"], plan.info["build_upload_s"]))
        if uniform and dtype == "float32":
            # the same table as COO segment rows in REVERSE order (wagg_dense_create_from_segments, 4 GB of host arrays):
            # nothing about the row order survives the sort -- the same plan again
            cell = np.repeat(np.arange(G, dtype=np.int32), np.diff(rowptr))[::-1].copy()
            t0 = time.time()
            coo = engine.DensePlan.from_segments(cell, col[::-1].copy(), val[::-1].copy(), G, R, dtype=dtype)
            t_coo = time.time() - t0
            del cell
            assert coo.info["form"] == plan.info["form"] and coo.info["nnz"] == len(col) and coo.info["w_bytes"] == plan.info["w_bytes"]
            np.testing.assert_array_equal(coo.den, plan.den)
            assert torch.equal(coo.apply(X[:256].contiguous()), plan.apply(X[:256].contiguous()))
            print("c5-uniform float32 as 2.5e8 COO rows in reverse order: from_segments %.2f s" % t_coo)
            coo.close()
        del got, want, X
        plan.close(); ref.close()
        torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------
# +-inf met by a REPLICA of an MFMA-form plan must reach the caller
# ---------------------------------------------------------------------------------------------
def test_inf_in_a_block_dealt_to_a_replica_is_not_lost(torch_cuda):
    """Multi-device host streaming of a dense-family plan (wagg_dense_apply_host_multi_*): row blocks are dealt round-robin
    to replica plans, each of which notes +-inf in its own word.  The notes are collected in plans[0] (and cleared in the
    replicas), so wagg_dense_saw_inf on the primary answers for the whole call and the drop-in redoes the field through the
    exact segment-table form: inf stays with the regions that own the cell (S6) instead of turning other regions NaN."""
    import pandas as pd
    from climate_toolbox_amd import _lib, aggregations as A, minixr
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    rng = np.random.default_rng(21)
    # library level: a full-matrix plan and one replica on the same device, +-inf only in rows of the SECOND block
    G, R, T = 32 * 1300, 300, 1500
    W = rng.uniform(0, 1, (G, R)).astype(np.float32)
    plan = DensePlan.from_host(W)
    rep = plan.replica(0)
    block_rows, n_blocks = _lib.host_block_plan(T, G * 4, 368, 2)
    assert n_blocks >= 2
    X = (280 + 10 * rng.standard_normal((T, G))).astype(np.float32)
    assert not plan.saw_inf() and not rep.saw_inf()
    plan.apply_host(X, flags=0, replicas=[rep])
    assert not plan.saw_inf()
    X[block_rows + 3, 77] = np.inf                       # block 1 -> pipeline 1 = the replica
    plan.apply_host(X, flags=0, replicas=[rep])
    assert plan.saw_inf() and not rep.saw_inf()          # collected in the primary, cleared in the replica
    assert not plan.saw_inf()                            # ... and reading clears it
    plan.close(); rep.close()
    # drop-in level: a scattered, dense-ish table takes an MFMA form (full matrix or all tiles stored); HOST_DEVICES = [0, 0]
    nlat, nlon, R, T = 24, 48, 1024, 2200                # (four whole column tiles of regions: the form choice goes by padded work
                                                         #  and by how many CUs a launch can fill)
    lat, lon = np.arange(nlat) * 1.0, np.arange(nlon) * 1.0
    n = int(0.3 * nlat * nlon * R)
    flat = rng.choice(nlat * nlon * R, size=n, replace=False)
    cell, lab = flat // R, flat % R
    df = pd.DataFrame({"lat": lat[cell // nlon], "lon": lon[cell % nlon], "areawt": rng.uniform(0.1, 1, n), "hierid": lab})
    tas = (280 + 10 * rng.standard_normal((T, nlat, nlon))).astype(np.float32)
    block_rows, n_blocks = _lib.host_block_plan(T, nlat * nlon * 4, 368, 2)
    assert n_blocks >= 4
    tas[block_rows + 5, 5, 7] = np.inf                   # a row of block 1 (the replica's)
    tas[3 * block_rows + 1, 9, 1] = -np.inf              # ... and of block 3
    ds = minixr.Dataset({"tas": (("time", "lat", "lon"), tas)}, coords={"time": np.arange(T), "lat": lat, "lon": lon})
    saved = A.HOST_DEVICES
    try:
        A.HOST_DEVICES = [0, 0]
        A._PLAN_CACHE.clear()
        out = A.weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df).tas.values
        (plan,) = [p for p in A._PLAN_CACHE.values() if isinstance(p, DensePlan)]
        assert plan.info["form"] in (_lib.FORM_FULL, _lib.FORM_TILES) and len(plan._replicas) == 1         # an MFMA form
    finally:
        A.HOST_DEVICES = saved
    args = (("time", "lat", "lon"), lat, lon, df["lat"].values, df["lon"].values, df["areawt"].values, df["areawt"].values, df["hierid"].values)
    ref = O.agg_scatter(tas, *args, group_dim="hierid")[0]
    assert np.isinf(ref).any() and not np.isnan(ref).any()
    assert not np.isnan(out).any()
    _rel_ok(out, ref, RTOL32)
    # the default keeps host fields on the current device: no replicas are built
    A._PLAN_CACHE.clear()
    A.weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
    assert all(not getattr(p, "_replicas", {}) for p in A._PLAN_CACHE.values())


# ---------------------------------------------------------------------------------------------
# device-resident fields of other dtypes, take_axis corner cases
# ---------------------------------------------------------------------------------------------
def test_device_fields_of_other_dtypes_are_promoted_by_the_library(torch_cuda):
    """A device-resident variable that is not float32/float64 is promoted to float64 like the reference promotes it (data
    times float64 weights, S8) -- by wagg_relayout_to_f64, strided sources included; take_axis accepts negative axes, host
    tensors and narrow dtypes."""
    from climate_toolbox_amd import engine, minixr, synth, weighted_aggregate_grid_to_regions
    torch = torch_cuda
    rng = np.random.default_rng(2)
    a = rng.integers(-100, 100, (5, 12, 9))
    for tdt, ndt in ((torch.int16, np.int16), (torch.int32, np.int32), (torch.int64, np.int64), (torch.uint8, np.uint8),
                     (torch.int8, np.int8), (torch.float16, np.float16), (torch.bfloat16, None), (torch.float32, np.float32)):
        t = torch.from_numpy(a.astype(np.float32)).to(tdt).cuda() if ndt is None else torch.from_numpy(a.astype(ndt)).cuda()
        want = t.cpu().to(torch.float64).numpy()
        np.testing.assert_array_equal(engine.to_float64(t).cpu().numpy(), want)
        np.testing.assert_array_equal(engine.to_float64(t[:, ::2, 1:7], order=(2, 0, 1)).cpu().numpy(),
                                      np.transpose(want[:, ::2, 1:7], (2, 0, 1)))
    t16 = torch.from_numpy(a.astype(np.int16)).cuda()
    idx = [8, 0, 3]
    np.testing.assert_array_equal(engine.take_axis(t16, -1, idx).cpu().numpy(), np.take(a, idx, axis=2).astype(np.float64))
    np.testing.assert_array_equal(engine.take_axis(torch.from_numpy(a), 1, [4, 4, 0]).numpy(), np.take(a, [4, 4, 0], axis=1))
    f = torch.from_numpy(a.astype(np.float32)).cuda()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = engine.take_axis(f, -2, [1, 11], stream=side)
    side.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), np.take(a, [1, 11], axis=1).astype(np.float32))
    # the drop-in on an int16 device field = the drop-in on the same numbers as a float64 host array
    nlat, nlon, T = 48, 96, 12
    lat, lon, df = synth.realistic_segments(nlat, nlon, R=40, seed=4, string_labels=False)
    vals = rng.integers(250, 320, (T, nlat, nlon)).astype(np.int16)
    outs = []
    for v in (vals.astype(np.float64), torch.from_numpy(vals).cuda()):
        ds = minixr.Dataset({"tas": (("time", "lat", "lon"), v)}, coords={"time": np.arange(T), "lat": lat, "lon": lon})
        outs.append(weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df).tas.values)
    assert outs[1].dtype == np.float64
    np.testing.assert_allclose(outs[1], outs[0], rtol=1e-12)


# ---------------------------------------------------------------------------------------------
# RCCL on the hardware at hand: one rank, backend "nccl", device tensors
# ---------------------------------------------------------------------------------------------
def test_bench_over_rccl_with_one_rank():
    """bench.py as a FRESH child process with WAGG_BENCH_FORCE_DIST=1: RCCL initialises on the GPU, every timed step
    queues ShardedStep's asynchronous dist.gather of a device tensor into row views of the destination, and the line says
    the reassembled series is right.  Still one rank: what it proves is that RCCL accepts the calls, not how they scale."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WAGG_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29531")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2-real", "--steps", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    from tests.bench_io import split_bench_output
    line, rec = split_bench_output(p)              # one stdout line under 6 KB; `rec` = the detail from stderr
    assert line["roofline"]["frac"] > 0 and line["rccl_ranks"] == 1 and line["gather_ok"] is True and abs(line["value"] / rec["value"] - 1) < 1e-5
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["gather_ok"] is True and rec["scaling_measured"] is False
    assert rec["roofline"]["bound"] == "hbm" and rec["value"] > 0 and rec["min_ms"] <= rec["median_ms"] <= rec["max_ms"]
    # the N > 1 fields, here over RCCL itself (one rank): the all-reduce proof, one blocking gather timed, per-rank medians
    assert rec["rccl_ranks"] == 1 and rec["backend"] == "nccl" and rec["gather_ms"] > 0 and len(rec["per_rank_ms"]) == 1
    rf = rec["roofline"]                   # kernel timings: hipExtLaunchKernel start / stop events, median-based fraction
    assert rf["kernel_launches"] == 3 and rf["kernel_ms_min"] <= rf["kernel_ms_median"] <= rf["kernel_ms_max"] <= rec["max_ms"]
    assert rf["frac_basis"] == "median kernel time" and rec["warmup_done"] >= rec["warmup"]


_RCCL_CHILD = r"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["WAGG_ROOT"])
from climate_toolbox_amd.timeshard import ShardedStep, gather_time_shards
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
T, R = 37, 1000
blk = torch.arange(T * R, dtype=torch.float32, device="cuda").reshape(T, R)
# the ragged route (grouped point-to-point into row views) and the equal route (dist.gather into row views), both async
for rows in ([T], None):
    dst = torch.full((T, R), -1.0, device="cuda")
    res, h = gather_time_shards(blk, dst=0, rows=rows, out=dst, async_op=True)
    h.wait(); torch.cuda.synchronize()
    assert res is dst and torch.equal(dst, blk), rows
    assert torch.equal(gather_time_shards(blk * 2, rows=rows), blk * 2)
k = [0]
def apply(out):
    k[0] += 1
    out.fill_(float(k[0]))
st = ShardedStep(apply, lambda: torch.empty((T, R), device="cuda"), rows=[T], dst=0, distributed=True)
for _ in range(5):
    st.step()
got = st.finish(); torch.cuda.synchronize()
assert bool((got == 5.0).all())
dist.barrier(); dist.destroy_process_group()
print("RCCL-OK")
"""


def test_time_shard_gather_over_rccl_with_one_rank():
    """climate_toolbox_amd/timeshard.py on backend "nccl" (= RCCL) with device tensors, one rank, in a fresh child process:
    the ragged route, the equal route and the double-buffered ShardedStep."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WAGG_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _RCCL_CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "RCCL-OK" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])


# ---------------------------------------------------------------------------------------------
# prepare_weights: the table coded once
# ---------------------------------------------------------------------------------------------
def test_prepared_weights_give_the_dataframe_results(torch_cuda, tmp_path):
    """weights=prepare_weights(df, ...) through the three reference-named functions, the fused tas_poly and the CSV path
    route: the same bits as the DataFrame (which is fingerprinted by content on every call), one plan for both, no
    re-hashing of the table columns on the prepared route."""
    import pandas as pd
    from climate_toolbox_amd import (aggregations as A, minixr, prepare_weights, synth, tas_poly_aggregate,
                                     weighted_aggregate_grid_to_regions, _aggregate_reindexed_data_to_regions,
                                     _reindex_spatial_data_to_regions)
    torch = torch_cuda
    nlat, nlon, T = 96, 192, 40
    lat, lon, df = synth.realistic_segments(nlat, nlon, R=150, seed=6, string_labels=True)
    df["popwt"] = np.random.default_rng(1).lognormal(0, 1, len(df))
    df.loc[::5, "popwt"] = np.nan
    rng = np.random.default_rng(3)
    tas = (280 + 10 * rng.standard_normal((T, nlat, nlon))).astype(np.float32)
    tas[3, 4, 5] = np.nan
    for dev in (False, True):
        v = torch.from_numpy(tas).cuda() if dev else tas
        ds = minixr.Dataset({"tas": (("time", "lat", "lon"), v)},
                            coords={"time": pd.date_range("2001-01-01", periods=T).values, "lat": lat, "lon": lon})
        A._PLAN_CACHE.clear()
        want = weighted_aggregate_grid_to_regions(ds, "tas", "popwt", "hierid", df)
        prep = prepare_weights(df, "popwt", "hierid")
        calls = {"n": 0}
        real = A._fingerprint

        def counting(*a, **k):
            calls["n"] += 1
            return real(*a, **k)

        A._fingerprint = counting
        try:
            got = weighted_aggregate_grid_to_regions(ds, "tas", "popwt", "hierid", prep)
            first = calls["n"]
            got2 = weighted_aggregate_grid_to_regions(ds, "tas", "popwt", "hierid", prep)
            assert calls["n"] == first                                   # second call: nothing hashed at all
        finally:
            A._fingerprint = real
        assert len(A._PLAN_CACHE) == 1                                    # the DataFrame and its snapshot share the plan
        for g in (got, got2):
            np.testing.assert_array_equal(g.tas.values, want.tas.values)
            assert list(g["hierid"].values) == list(want["hierid"].values)
        re = _reindex_spatial_data_to_regions(ds, prep)
        np.testing.assert_array_equal(_aggregate_reindexed_data_to_regions(re, "tas", "popwt", "hierid", prep).tas.values, want.tas.values)
        with pytest.raises(ValueError, match="prepared for"):
            weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", prep)
        p1 = tas_poly_aggregate(ds, [1, 2], "popwt", "hierid", df)
        p2 = tas_poly_aggregate(ds, [1, 2], "popwt", "hierid", prep)
        for k in ("tas-poly-1", "tas-poly-2"):
            np.testing.assert_array_equal(p2[k].values, p1[k].values)
    # the CSV route: memoised on the path (aggregations.py:127), coded once per (path, aggwt, agglev)
    csv = df.rename(columns={"lon": "pix_cent_x", "lat": "pix_cent_y"})
    path = str(tmp_path / "weights.csv")
    csv.to_csv(path, index=False)
    ds = minixr.Dataset({"tas": (("time", "lat", "lon"), tas)}, coords={"time": np.arange(T), "lat": lat, "lon": lon})
    a = weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", path)
    assert (path, "areawt", "hierid") in A._PREPARED_BY_PATH
    b = weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", path)
    np.testing.assert_array_equal(a.tas.values, b.tas.values)
    np.testing.assert_allclose(a.tas.values, weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df).tas.values, rtol=1e-6)


@pytest.mark.parametrize("dtype,rtol", [(np.float32, RTOL32), (np.float64, 1e-11)])
def test_table_builder_corner_shapes_vs_oracle(torch_cuda, dtype, rtol):
    """Shapes at the edges of the device builder's geometry, each against the oracle's aggregation of the same rows: one
    grid cell; one region; every row in ONE cell; one pair repeated 5,000 times (a single run for the coalescing kernel);
    a region whose weights cancel to 0 and one that is all zeros (S7: inf / NaN); negative weights; G and R that are not
    multiples of anything (chunks of 128 cells, 16-wave region blocks, 256-region tiles); a table larger than one sort
    tile per wave (> 8,192 rows) with most rows dropped (null labels), so the dropped keys cross tile borders."""
    from climate_toolbox_amd.engine import DensePlan
    from oracle import ref_numpy as O
    torch = torch_cuda
    rng = np.random.default_rng(99)
    cases = []
    cases.append(("one cell", 1, 7, np.zeros(30, np.int32), rng.integers(0, 7, 30).astype(np.int32), rng.uniform(0.1, 1, 30)))
    cases.append(("one region", 777, 1, rng.integers(0, 777, 400).astype(np.int32), np.zeros(400, np.int32), rng.uniform(0.1, 1, 400)))
    cases.append(("one cell of many", 1000, 50, np.full(300, 613, np.int32), rng.integers(0, 50, 300).astype(np.int32), rng.uniform(0.1, 1, 300)))
    cases.append(("one pair 5000 times", 130, 20, np.full(5000, 129, np.int32), np.full(5000, 19, np.int32), rng.uniform(0.1, 1, 5000)))
    w = rng.uniform(0.1, 1, 200)
    code = rng.integers(2, 9, 200).astype(np.int32)
    cell = rng.integers(0, 257, 200).astype(np.int32)
    cell = np.concatenate([cell, [5, 6, 7, 8]]).astype(np.int32)
    code = np.concatenate([code, [0, 0, 1, 1]]).astype(np.int32)        # region 0: +2 - 2 = 0 (x / 0 = +-inf, or NaN); region 1: 0 + 0
    w = np.concatenate([w, [2.0, -2.0, 0.0, 0.0]])
    w[::17] *= -1.0                                                      # negative weights are weights (the backup fill is upstream)
    cases.append(("zero denominators, negative weights", 257, 9, cell, code, w))
    n = 40000
    code = rng.integers(0, 1201, n).astype(np.int32)
    code[rng.random(n) < 0.9] = -1                                       # nine rows in ten carry no label
    cases.append(("mostly dropped rows", 129 * 7 + 1, 1201, rng.integers(0, 129 * 7 + 1, n).astype(np.int32), code, rng.uniform(0.1, 1, n)))
    for name, G, R, cell, code, w in cases:
        T = 33
        X = (5 + rng.standard_normal((T, G))).astype(dtype)
        with np.errstate(divide="ignore", invalid="ignore"):
            ref = O.agg_coded(X, cell, code, w, R)
        Xd = torch.from_numpy(X).cuda()
        seg = DensePlan.from_segments(cell, code, w, G, R, dtype=dtype)
        got = seg.apply(Xd).cpu().numpy()
        keep = (code >= 0) & ~np.isnan(w)
        assert seg.info["nnz"] == len(np.unique(cell[keep].astype(np.int64) * R + code[keep])), name
        np.testing.assert_allclose(seg.den, np.bincount(code[keep], weights=w[keep], minlength=R), rtol=1e-12, atol=1e-15, err_msg=name)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)]), name
        _rel_ok(np.where(fin, got, 0.0), np.where(fin, ref, 0.0), rtol)
        order = np.argsort(cell, kind="stable")
        rowptr = np.concatenate([[0], np.cumsum(np.bincount(cell, minlength=G))]).astype(np.int64)
        csr = DensePlan.from_csr(rowptr, code[order], w[order], G, R, dtype=dtype)
        np.testing.assert_array_equal(csr.apply(Xd).cpu().numpy(), got, err_msg=name)
        seg.close(); csr.close()


def test_bench_multi_rank_control_flow_rehearsal():
    """`bench.py --gpus 3` with WAGG_BENCH_REHEARSE=gloo (every rank on device 0, blocks gathered through host memory): the
    N > 1 control flow the driver's scaling run takes -- the weak-scaled headline, then the STRONG splits of configs[3] and
    configs[4] as `secondary` (equal and ragged row counts), every gather checked -- on a small grid.  Control flow only:
    the line says it is not a measurement, and RCCL is covered by the one-rank tests above."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WAGG_BENCH_REHEARSE="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--nlat", "180", "--nlon", "360", "--R", "600",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    from tests.bench_io import split_bench_output
    line, rec = split_bench_output(p)              # one stdout line under 6 KB; `rec` = the detail from stderr
    # the compact line is what the driver's scaling run parses: headline + roofline + the N > 1 proof + one row per workload
    assert line["n_gpus"] == 3 and line["roofline"]["frac"] > 0 and line["rccl_ranks"] == 3 and len(line["per_rank_ms"]) == 3
    assert line["gather_ms"] > 0 and line["gather_ok"] is True and "cpu_baseline" in line
    assert [r["wl"] for r in line["summary"]["rows"]] == ["c2-dense", "c4", "c5-uniform"]
    assert all(r["gather_ok"] is True and r["gather_ms"] > 0 for r in line["summary"]["rows"])
    assert rec["n_gpus"] == 3 and rec["scaling"] == "weak" and rec["scaling_measured"] is False and "rehearsal" in rec
    assert rec["gather_ok"] is True and rec["config"]["T_job"] == 3 * 365
    # what the driver needs to check an N > 1 line: the all-reduce-of-ones proof that the collective
    # backend saw every rank, one blocking gather of the real block timed (max over ranks), every rank's median step
    assert rec["rccl_ranks"] == 3 and rec["backend"] == "gloo" and rec["gather_ms"] > 0 and len(rec["per_rank_ms"]) == 3
    assert rec["gather_bytes"] == 3 * 365 * 600 * 4 and all(ms > 0 for ms in rec["per_rank_ms"])
    assert rec["summary"]["rows"][0]["wl"] == "c2-dense" and [r["wl"] for r in rec["summary"]["rows"][1:]] == ["c4", "c5-uniform"]
    sec = {s["workload"]: s for s in rec["secondary"]}
    assert set(sec) == {"c4", "c5-uniform"}
    assert sec["c4"]["scaling"] == "strong" and sec["c4"]["T_job"] == 10950 and sec["c4"]["gather_ok"] is True
    assert sec["c5-uniform"]["scaling"] == "strong" and sec["c5-uniform"]["T_job"] == 18250 and sec["c5-uniform"]["gather_ok"] is True
    assert sec["c5-uniform"]["T"] == 6084                                # ragged: 6084 + 6083 + 6083 rows
    for wl in ("c4", "c5-uniform"):
        assert sec[wl]["rccl_ranks"] == 3 and sec[wl]["gather_ms"] > 0 and len(sec[wl]["per_rank_ms"]) == 3
    # a spent wall-clock budget drops the secondaries instead of running into the driver's limit
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--nlat", "180", "--nlon", "360", "--R", "600",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--budget-s", "0"], capture_output=True, text=True, env=env,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line, rec = split_bench_output(p)
    assert [(r["wl"], r.get("skipped")) for r in line["summary"]["rows"][1:]] == [("c4", "budget"), ("c5-uniform", "budget")]
    assert [(s_["workload"], s_["skipped"]) for s_ in rec["secondary"]] == [("c4", "budget"), ("c5-uniform", "budget")]


def test_integration_md_csr_binding_runs_as_written(torch_cuda):
    """The ctypes stub INTEGRATION.md shows for a table that is already CSR is executed as written (in the namespace of the
    stub of section 2), on a scipy.sparse matrix, and its plan applied through wagg_dense_apply_f32."""
    import ctypes as C
    import re
    import scipy.sparse as sp
    from climate_toolbox_amd import _lib
    from oracle import ref_numpy as O
    torch = torch_cuda
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    base = re.search(r"```python\n(# climate_toolbox/aggregations/_wagg.py.*?)```", text, re.S).group(1)
    base = base.replace('C.CDLL("libwagg.so")', "C.CDLL(%r)" % _lib.LIB_PATH)
    block = re.search(r"```python\n(def plan_from_csr.*?)```", text, re.S).group(1)
    L = _lib.load()
    ns = {}
    exec(compile(base, "INTEGRATION.md", "exec"), ns)
    exec(compile(block, "INTEGRATION.md#csr", "exec"), ns)
    rng = np.random.default_rng(5)
    G, R, T = 3000, 120, 40
    W = sp.random(G, R, density=0.03, format="csr", random_state=7, data_rvs=lambda k: rng.uniform(0.1, 1, k))
    h = ns["plan_from_csr"](W.indptr.astype(np.int64), W.indices.astype(np.int32), W.data.astype(np.float64), G, R)
    try:
        X = (280 + 10 * rng.standard_normal((T, G))).astype(np.float32)
        Xd = torch.from_numpy(X).cuda()
        out = torch.empty((T, R), dtype=torch.float32, device="cuda")
        rc = L.wagg_dense_apply_f32(h, C.c_void_p(Xd.data_ptr()), T, G, C.c_void_p(out.data_ptr()), R, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, L.wagg_last_error()
        torch.cuda.synchronize()
    finally:
        L.wagg_dense_destroy(h)
    coo = W.tocoo()
    _rel_ok(out.cpu().numpy(), O.agg_coded(X, coo.row.astype(np.int32), coo.col.astype(np.int32), coo.data, R), RTOL32)
