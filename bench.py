#!/usr/bin/env python3
"""bench.py -- gridcell-region-timesteps/s of the weighted grid->region aggregation on MI355X.

A "step" is one pass of the hot path over one batch: out[T, R] = (X[T, G] . W[G, R]) / den[R].

Default workload (N=1) = BASELINE.json configs[1] in its dense north-star form ("c2-dense"):
T=365 daily steps, 0.25-degree grid G=720x1440=1,036,800, R=24,378 impact regions, fp32, W a
dense (gridcell x region) matrix generated on the device (101 GB, hash of (g, r, seed)); inputs
are resident in HBM before the timed region.  The same run also times "c2-real" / "c3-real" (the
sparse segment-table form of configs[1] / [2]) and reports them under "secondary".

Workloads (--workload):
  c1          configs[0]: 2-degree grid, 100 regions, fp64 -- the reference's own CPU-sized case; the
              CPU oracle is timed on the FULL problem beside it
  c2-dense    configs[1], dense north-star form (default)
  c2-real     configs[1], segment-table form (HBM-bound)     c3-real  configs[2], fp64 + popwt
  c4          configs[3]: 10,950 daily steps, dense form, time axis cut over the ranks ("strong")
  c5-uniform  configs[4]: 50 x 365 rows, ~1 % non-zeros at uniformly random positions (entry-list form)
  c5-block    configs[4]: the block-local structure (tile-sparse MFMA form); c5-block-f64: the same in fp64
              (v_mfma_f64_16x16x4_f64)
For c4/c5 the rows of a run are shard_bounds(T_total, --shards)[rank]; --shards defaults to the
world size for c4 and to 8 for c5 (one rank's share of the 8-GPU job on a single GPU).

With --gpus N the time axis is sharded over N ranks (one process per GPU) and the region time series are
reassembled on rank 0 with an RCCL gather that overlaps the next step's compute; the default workload is weak
scaling (365 rows per rank).  The ranks may be started by torch.distributed.run (the driver's way: RANK /
LOCAL_RANK / WORLD_SIZE in the environment) or by this script itself: a plain `python bench.py --gpus N` starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child BEFORE anything touches the GPU,
relays rank 0's JSON line and returns the child's exit status (a rank whose LOCAL_RANK has no device exits
non-zero with a one-line reason).  --dry-run rehearses exactly that plumbing on CPU (gloo, no kernels).

The default N=1 run carries every BASELINE config in its line: c2-dense is the headline; "secondary" holds c1,
c2-real, c3-real, one rank's share of c4 (1,369 rows on the c2-dense operand) and of c5 in both structures
(2,282 rows; block-local fp32, uniform fp32 and fp64), each with its own roofline and cpu_baseline.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 flop/clk/CU
PEAK_F64_MFMA_TFLOPS = 78.6     # half the fp32 rate
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
WORKLOADS = ["c1", "c2-dense", "c2-real", "c3-real", "c4", "c5-block", "c5-uniform", "c5-block-f64", "c5-uniform-f64"]
DENSE_FAMILY = ("c2-dense", "c4", "c5-block", "c5-uniform", "c5-block-f64", "c5-uniform-f64")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2-dense", choices=WORKLOADS)
    ap.add_argument("--T", type=int, default=0, help="rows per rank (overrides the workload's own)")
    ap.add_argument("--shards", type=int, default=0, help="c4/c5: cut T_total into this many shards, this rank takes its own")
    ap.add_argument("--nlat", type=int, default=720)
    ap.add_argument("--nlon", type=int, default=1440)
    ap.add_argument("--R", type=int, default=24378)
    ap.add_argument("--ksplit", type=int, default=0)
    ap.add_argument("--land-frac", type=float, default=0.30, help="experiment knob for c2-real")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the launcher and the rank plumbing on CPU (gloo, no kernels, value 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--diag-lib", nargs="?", const="libwagg_diag.so", default=None, metavar="NAME",
                    help="load climate_toolbox_amd/lib/NAME (default libwagg_diag.so, `make diag`) instead of libwagg.so: "
                         "the build with the ablation knobs, or an experiment's variant; timing experiments only")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` without torch.distributed.run in front: start the N ranks as a child job and relay
    rank 0's line.  Nothing here touches the GPU (no torch import, no libwagg call): the children are fresh
    processes, this one only waits."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:                 # the ranks' stderr passes straight through
        if ln.startswith("{"):
            line = ln
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited cleanly without a JSON line\n")
        rc = 1
    return rc


def dry_run(a, rank, local_rank, world):
    """The rank plumbing without a GPU: gloo group, the shard rule, barrier + max-over-ranks timing, one line."""
    import torch
    import torch.distributed as dist
    from climate_toolbox_amd.timeshard import ShardedStep, shard_bounds
    if world > 1:
        dist.init_process_group("gloo")
    rows = [e - s0 for s0, e in shard_bounds(365 * world, world)]
    R = 16
    st = ShardedStep(lambda out: out.fill_(float(rank + 1)), lambda: torch.empty((rows[rank], R)), rows=rows, dst=0,
                     distributed=world > 1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st.step()
    got = st.finish()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        ok = world == 1 or all(bool((got[sum(rows[:r]):sum(rows[:r + 1])] == r + 1).all()) for r in range(world))
        print(json.dumps({"metric": "gridcell-region-timesteps/sec", "value": 0.0, "unit": "gridcell-region-timesteps/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / max(1, a.steps) * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "dry_run": True, "gather_ok": ok,
                          "config": {"workload": "dry run: launcher + gloo gather only, no kernels", "T_job": sum(rows)}}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def load_traffic(workload):
    """(HBM bytes per launch, where that number comes from).  It is NOT measured by this run: it is
    the figure tools/summarize_prof.py derived from separate rocprofv3 --pmc passes and committed to
    profiles/traffic.json (PMC collection needs its own profiler runs, see profiles/README.md)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as f:
            e = json.load(f).get(workload)
        if not e:
            return None, None
        return e.get("hbm_bytes_per_launch"), "profiles/traffic.json <- %s (%s)" % (e.get("source"), e.get("measured", "round 1"))
    except Exception:
        return None, None


# wagg_plan_info.kernel_form (when the library reports it): the dominant kernel of a segment-table apply
PLAN_KERNEL = {}


def sparse_algorithmic_bytes(T, G, R, nnz, b):
    # b*T*G + (b+4)*nnz + 4*(G+1) + b*T*R + b*R
    return b * T * G + (b + 4) * nnz + 4 * (G + 1) + b * T * R + b * R


def timed_steps(torch, dist, step, finish, steps, warmup, world):
    """(wall seconds of exactly `steps` steps between barrier + synchronize on both sides, max over ranks;
    per-step milliseconds from event pairs on the compute stream -- recorded without any synchronisation)."""
    world = world if not (dist.is_available() and dist.is_initialized()) else max(world, 2)   # forced-dist rehearsal
    for _ in range(warmup):
        step()
    finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a_, b_ in ev:
        a_.record()
        step()
        b_.record()
    finish()                               # every queued gather has completed
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, sorted(a_.elapsed_time(b_) for a_, b_ in ev)


def step_stats(per_step_ms):
    """median and minimum of the per-step device times (SURVEY 8d's timing protocol asks for both)."""
    return {"median_ms": per_step_ms[len(per_step_ms) // 2], "min_ms": per_step_ms[0]}


# ---- CPU baselines (rank 0, N = 1 only; bounded samples; the oracle is the thing timed here) -------
def cpu_baseline_dense(T, G, R, seed_w, fill=1.0, blocklocal=False):
    """The oracle's dense restatement (oracle/wagg_oracle.c, kind "port") on a bounded window of
    the same workload: Tw rows x Gw cells x Rw regions, threads = OpenMP default."""
    import numpy as np
    from oracle import c_oracle
    Tw, Gw, Rw = min(T, 365), min(G, 32768), min(R, 8192)
    rng = np.random.default_rng(0)
    X = (280.0 + 30.0 * rng.standard_normal((Tw, Gw))).astype(np.float32)
    c_oracle.dense_synth_sparse(X[:8], 0, min(Gw, 256), R, 0, min(Rw, 256), seed_w, fill, blocklocal)     # warm
    t0 = time.perf_counter()
    c_oracle.dense_synth_sparse(X, 0, Gw, R, 0, Rw, seed_w, fill, blocklocal)
    dt = time.perf_counter() - t0
    return {"value": Tw * Gw * Rw / dt, "unit": "gridcell-region-timesteps/s", "cores": c_oracle.threads(),
            "kind": "port", "wall_s": round(dt, 3),
            "sample": "oracle/wagg_oracle.c dense (fp64 accumulate, OpenMP; weights regenerated from the hashes, "
                      "fill=%g%s) on %d rows x %d cells x %d regions of this workload's operands"
                      % (fill, ", block-local" if blocklocal else "", Tw, Gw, Rw)}


def cpu_baseline_sparse(X_host, cell, codes, w_eff, R, G, what):
    """The oracle's faithful single-threaded restatement (gather -> fp64 multiply -> group-sum ->
    divide, like the reference) on the sample handed in (the FULL workload for the segment-table
    configs: it is only ~0.4 s of CPU work), median and minimum of up to 15 repeats inside ~10 s; plus the
    best-effort CPU leg SURVEY 8d asks for -- the same sums with the timesteps dealt to all host cores
    (OpenMP, oracle/wagg_oracle.c::wagg_oracle_segments_omp_*) -- and scipy's CSC SpMM (one thread) for reference."""
    import numpy as np
    import scipy.sparse as sp
    from oracle import c_oracle
    Ts = X_host.shape[0]
    c_oracle.segments(X_host[:1], cell, codes, w_eff, R)

    def timed(fn, budget_s=10.0, max_reps=15):
        ts, t_start = [], time.perf_counter()
        while len(ts) < max_reps and (len(ts) < 3 or time.perf_counter() - t_start < budget_s):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s and len(ts) >= 1:
                break
        ts.sort()
        return ts[len(ts) // 2], ts[0], len(ts)

    dt, dmin, reps = timed(lambda: c_oracle.segments(X_host, cell, codes, w_eff, R))
    c_oracle.segments(X_host[:c_oracle.threads()], cell, codes, w_eff, R, threaded=True)          # (thread pool start-up)
    dto, domin, repso = timed(lambda: c_oracle.segments(X_host, cell, codes, w_eff, R, threaded=True))
    keep = (codes >= 0) & ~np.isnan(w_eff)
    W = sp.coo_matrix((w_eff[keep], (cell[keep], codes[keep])), shape=(G, R)).tocsc()
    den = np.asarray(W.sum(axis=0)).ravel()

    def scipy_leg():
        Xs = np.nan_to_num(X_host.astype(np.float64), nan=0.0, posinf=np.inf, neginf=-np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            return (W.T @ Xs.T).T / den[None, :]

    dtb, dbmin, repsb = timed(scipy_leg, budget_s=5.0, max_reps=5)
    return {"value": Ts * G * R / dt, "unit": "gridcell-region-timesteps/s", "cores": 1, "kind": "port",
            "wall_s": round(dt, 4), "wall_s_min": round(dmin, 4), "repeats": reps, "nnz_timesteps_per_s": Ts * len(cell) / dt,
            "sample": "oracle/wagg_oracle.c segments_%s (single thread, like the reference), %s"
                      % ("f32" if X_host.dtype == np.float32 else "f64", what),
            "cpu_best": {"value": Ts * G * R / dto, "wall_s": round(dto, 4), "wall_s_min": round(domin, 4), "cores": c_oracle.threads(),
                         "repeats": repso, "kind": "port",
                         "what": "the same C restatement with the timesteps dealt to every host core (OpenMP), same sample"},
            "cpu_scipy": {"value": Ts * G * R / dtb, "wall_s": round(dtb, 4), "wall_s_min": round(dbmin, 4), "cores": 1, "repeats": repsb,
                          "what": "scipy.sparse CSC^T @ X^T in fp64 (one thread), same sample"}}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a)              # before anything that could initialise the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.dry_run:
        return dry_run(a, rank, local_rank, world)
    import numpy as np
    import torch
    import torch.distributed as dist
    n_dev = torch.cuda.device_count()      # (counting devices does not initialise the GPU)
    # WAGG_BENCH_REHEARSE=gloo: the N > 1 CONTROL FLOW of this script (strong splits, secondaries, ragged gathers, the line)
    # on a box with fewer GPUs than ranks: every rank computes on device 0, the blocks travel through host buffers over gloo.
    # Not a measurement -- the line says so -- and RCCL is not involved (tests/test_gpu_round4.py covers RCCL with one rank).
    rehearse = os.environ.get("WAGG_BENCH_REHEARSE") == "gloo" and world > 1
    if rehearse:
        local_rank = 0
    if local_rank >= n_dev:
        raise SystemExit("bench.py: rank %d (LOCAL_RANK %d) has no device: this box has %d GPU(s), --gpus %d"
                         % (rank, local_rank, n_dev, a.gpus))
    torch.cuda.set_device(local_rank)
    # WAGG_BENCH_FORCE_DIST=1 exercises the RCCL init + gather code path with a single rank (rehearsal
    # on a one-GPU box); the timed step then includes the (trivial) gather exactly like N > 1 runs
    force_dist = os.environ.get("WAGG_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        # stdout carries exactly one line (the JSON below): keep RCCL's version banner off it
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ.pop("NCCL_DEBUG")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from climate_toolbox_amd import _lib, engine, synth
    if a.diag_lib:
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.path.basename(a.diag_lib))
    from climate_toolbox_amd.timeshard import ShardedStep, shard_bounds

    G, R = a.nlat * a.nlon, a.R
    scaling = "weak"
    C5_TOTAL, C4_TOTAL = 50 * 365, 10950

    def rows_for(wl, share=False):
        """(rows of this rank, rows of every rank, scaling) of a workload in this run; ``share``: one rank's share of
        the 8-GPU job (the secondary entries of the default N = 1 run)."""
        if wl in ("c4",) or wl.startswith("c5"):
            T_total = C4_TOTAL if wl == "c4" else C5_TOTAL
            shards = a.shards or (8 if share else (world if wl == "c4" or world > 1 else 8))
            bounds = shard_bounds(T_total, max(shards, world))
            rows_all = [e - s0 for s0, e in bounds][:world]
            if a.T:
                rows_all = [a.T] * world
            return rows_all[rank], rows_all, ("strong" if shards == world and not a.T else "weak")
        Tn = a.T or 365
        return Tn, [Tn] * world, "weak"

    def stepper(apply_fn, Tn, rows_all, Rr, dtype):
        if rehearse:                         # gloo moves host memory: compute into a device block, hand a host copy to the gather
            dev_block = torch.empty((Tn, Rr), dtype=dtype, device="cuda")

            def apply_to_host(out_host):
                apply_fn(dev_block)
                out_host.copy_(dev_block)
            return ShardedStep(apply_to_host, lambda: torch.empty((Tn, Rr), dtype=dtype), rows=rows_all, dst=0, distributed=True)
        return ShardedStep(apply_fn, lambda: torch.empty((Tn, Rr), dtype=dtype, device="cuda"), rows=rows_all, dst=0,
                           distributed=use_dist)

    def gather_check(st, Tn):
        """rank 0, distributed runs: the reassembled series holds this rank's last block in this rank's rows (the other
        ranks' rows are theirs to vouch for: the CPU gloo tests compare every block)."""
        if not use_dist or rank != 0:
            return None
        got = st.finish()
        return bool(got is not None and got.shape[0] == sum(st.rows) and torch.equal(got[:Tn], st.bufs[(st.k - 1) & 1]))

    def kernel_avg_ms(kms, warmup):
        kms = kms[warmup:] if len(kms) > warmup else kms
        return sum(kms) / max(1, len(kms))

    def on_traffic(roof, kavg, peak_gbs=PEAK_HBM_GBS):
        """The same roofline priced on the bytes the counters saw move (profiles/traffic.json), next to the
        algorithmic figure: a kernel that skips cells it does not need shows the difference here."""
        if roof.get("traffic"):
            roof["achieved_on_traffic"] = roof["traffic"] / kavg / 1e9
            roof["frac_on_traffic"] = roof["achieved_on_traffic"] / peak_gbs
            if roof["traffic"] < roof.get("algorithmic_bytes_per_launch", 0):
                roof["note"] = ("`achieved`/`frac` price SURVEY 8d's algorithmic bytes (all of X once); this kernel reads only "
                                "the 128-byte lines that hold referenced cells, so the fraction on those bytes can exceed 1: "
                                "`frac_on_traffic` is the one that says how close to the HBM the kernel runs")
        return roof

    def boundary_legs(Xs, lat, lon, df, plan, T, Gs, Rr):
        """What the kernel numbers leave out, on c2-real: the PCIe-inclusive rate of a host-resident
        field, the reference-named function end to end (device- and host-resident variable, cached plan), and the country
        level of the same table (agglev = "ISO": the other level the reference's docstring names, aggregations.py:104-106)."""
        from climate_toolbox_amd import _lib as L_, aggregations as A, minixr, weighted_aggregate_grid_to_regions
        out = {}
        Xh = Xs.cpu().numpy()
        plan.apply_host(Xh, flags=L_.HOST_PIN)                       # (first call: staging pieces, registration path warm)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            plan.apply_host(Xh, flags=L_.HOST_PIN)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out["host_resident"] = {"ms_per_step": ts[1] * 1e3, "min_ms": ts[0] * 1e3, "h2d_gbs": Xh.nbytes / ts[1] / 1e9,
                                "what": "wagg_apply_host_f32 (row-block pipeline, arrays page-locked in place): X from host memory, result "
                                        "back to host memory; PCIe-bound, never the headline value"}

        def calls(ds, n=30, warm=15):
            # (15 warm-up calls: the first dozen calls after an idle spell run 2-3x slower on this platform whatever they do --
            #  host cores and the copy engines come out of their idle states; tools/dropin_split_timing.py shows the switch)
            for _ in range(warm):
                weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
            ts = []
            for _ in range(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                weighted_aggregate_grid_to_regions(ds, "tas", "areawt", "hierid", df)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            return {"median_ms": ts[len(ts) // 2], "min_ms": ts[0], "calls": n}

        shape = (T, len(lat), len(lon))
        A._PLAN_CACHE.clear()
        dsd = minixr.Dataset({"tas": (("time", "lat", "lon"), Xs.reshape(shape))}, coords={"lat": lat, "lon": lon})
        dsh = minixr.Dataset({"tas": (("time", "lat", "lon"), Xh.reshape(shape))}, coords={"lat": lat, "lon": lon})
        from climate_toolbox_amd import prepare_weights
        df_plain = df
        prep = prepare_weights(df, "areawt", "hierid", lat=lat, lon=lon)
        d_dev, d_host = calls(dsd), calls(dsh, n=10, warm=5)
        df = prep                                                  # (calls() reads `df` of this scope)
        p_dev = calls(dsd)
        df = df_plain
        out["dropin_ms"] = {"device_resident": d_dev, "host_resident": d_host, "device_resident_prepared_weights": p_dev,
                            "what": "weighted_aggregate_grid_to_regions(ds, 'tas', 'areawt', 'hierid', df) end to end (aggregations.py:87), "
                                    "cached plan, result returned as a host array; c2-real table (%d rows) as a DataFrame (fingerprinted "
                                    "by content on every call) and as prepare_weights(df, ...) (coded once)" % len(df_plain)}
        A._PLAN_CACHE.clear()
        del dsd, dsh, Xh
        cell, codes, w_eff, uniq = synth.code_segments(df, lat, lon, "areawt", "ISO")
        t0 = time.perf_counter()
        iso = engine.SparsePlan(cell, codes, w_eff, Gs, len(uniq), row_len=len(lon))
        tb = time.perf_counter() - t0
        o = torch.empty((T, len(uniq)), dtype=Xs.dtype, device="cuda")
        for _ in range(3):
            iso.apply(Xs, out=o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            iso.apply(Xs, out=o)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        out["agglev_ISO"] = {"R": len(uniq), "ms_per_step": ms, "value": T * Gs * len(uniq) / (ms * 1e-3),
                             "unit": "gridcell-region-timesteps/s", "plan_build_s": round(tb, 4), "n_giant": int(iso.info["n_giant"]),
                             "what": "the same segment table aggregated to the country level (regions of thousands of cells each)"}
        iso.close()
        return out

    def run_sparse(dtype, small=False, steps=None, warmup=None, extras=True):
        steps, warmup = steps or a.steps, a.warmup if warmup is None else warmup
        T, rows_all, _ = rows_for("c1" if small else "c2-real")
        T_job = sum(rows_all)
        if small:       # c1: the reference's 2-degree test grid, 100 block regions
            lat, lon, tas, df = synth.c1_workload(T=T)
            Gs, nlon = len(lat) * len(lon), len(lon)
            wname, lev = "areawt", "hierid"
            Xs = torch.from_numpy(np.ascontiguousarray(tas.reshape(T, Gs))).cuda()
        else:
            lat, lon, df = synth.realistic_segments(nlat=a.nlat, nlon=a.nlon, R=R, seed=2, land_frac=a.land_frac)
            Gs, nlon = G, a.nlon
            wname, lev = ("areawt" if dtype == "float32" else "popwt"), "hierid"
            Xs = engine.synth_field(T, Gs, seed=1000 + rank, base=280.0, amp=60.0, dtype=dtype)
        cell, codes, w_eff, uniq = synth.code_segments(df, lat, lon, wname, lev)
        t0 = time.perf_counter()
        plan = engine.SparsePlan(cell, codes, w_eff, Gs, len(uniq), row_len=nlon)
        plan_build_s = time.perf_counter() - t0
        Rr = len(uniq)
        st = stepper(lambda out: plan.apply(Xs, out=out), T, rows_all, Rr, Xs.dtype)
        engine.profile_enable(True)       # event records only, no synchronisation
        dt, per_step = timed_steps(torch, dist, st.step, st.finish, steps, warmup, world)
        kavg = kernel_avg_ms(engine.profile_read(), warmup) * 1e-3
        engine.profile_enable(False)
        gok = gather_check(st, T)
        plan.status()
        b = 4 if dtype == "float32" else 8
        nnz = plan.info["nnz"]
        abytes = sparse_algorithmic_bytes(T, Gs, Rr, nnz, b)
        wl = "c1" if small else ("c2-real" if dtype == "float32" else "c3-real")
        traffic, tsrc = load_traffic(wl)
        kname = PLAN_KERNEL.get(int(plan.info.get("kernel_form", -1)),
                                "sparse_lcv_kernel<float>" if b == 4 else "sparse_lcv_kernel<double>")
        res = {
            "workload": wl, "dtype": "f32" if b == 4 else "f64", "T": T, "T_job": T_job, "G": Gs, "R": Rr, "nnz": int(nnz),
            "value": T_job * Gs * Rr * steps / dt, "unit": "gridcell-region-timesteps/s", "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, **step_stats(per_step), "nnz_timesteps_per_s": T_job * nnz * steps / dt,
            **({"gather_ok": gok} if gok is not None else {}),
            "plan_build_s": round(plan_build_s, 4),      # wagg_plan_create on the coded table (host-side chunking + upload), once per table
            "plan": {k: int(v) for k, v in plan.info.items()},
            "roofline": on_traffic({"bound": "hbm", "achieved": abytes / kavg / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": abytes / kavg / 1e9 / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": tsrc,
                                    "kernel": kname, "kernel_ms_avg": kavg * 1e3, "algorithmic_bytes_per_launch": abytes},
                                   kavg),
        }
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_sparse(Xs.cpu().numpy(), cell, codes, w_eff, Rr, Gs,
                                                      "the FULL workload (all %d timesteps, full segment table)" % T)
        if world == 1 and not small and extras:
            # the same field in (gridcell, time) order -- the reference's (lat, lon, time) fixture layout
            XT = Xs.t().contiguous()
            gout = torch.empty((T, Rr), dtype=Xs.dtype, device="cuda")
            for _ in range(3):
                plan.apply(XT, layout="GT", out=gout)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                plan.apply(XT, layout="GT", out=gout)
            torch.cuda.synchronize()
            res["layout_gridcell_time"] = {"ms_per_step": (time.perf_counter() - t0) / steps * 1e3}
            del XT, gout
            # fused tas_poly: (tas - 273.15)^p, p = 1..4, one pass over the field
            pout = torch.empty((4, T, Rr), dtype=Xs.dtype, device="cuda")
            for _ in range(2):
                plan.apply_poly(Xs, -273.15, 4, out=pout)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                plan.apply_poly(Xs, -273.15, 4, out=pout)
            torch.cuda.synchronize()
            pdt = (time.perf_counter() - t0) / steps
            res["fused_tas_poly_1to4"] = {"ms_per_step": pdt * 1e3, "value": 4 * T * Gs * Rr / pdt,
                                          "unit": "gridcell-region-timesteps/s (4 powers)"}
            del pout
            # fused Snyder degree days: one pass over (tasmin, tasmax) for 1 and for 3 thresholds
            tmax = Xs + engine.synth_field(T, Gs, seed=2000 + rank, base=6.0, amp=10.0, dtype=dtype)
            edd = {}
            for K in (1, 3):
                thr = [10.0, 20.0, 30.0][:K]
                eout = torch.empty((K, T, Rr), dtype=Xs.dtype, device="cuda")
                for _ in range(2):
                    plan.apply_edd(Xs, tmax, thr, offset=-273.15, out=eout)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    plan.apply_edd(Xs, tmax, thr, offset=-273.15, out=eout)
                torch.cuda.synchronize()
                edd["K%d_ms" % K] = (time.perf_counter() - t0) / steps * 1e3
                del eout
            res["fused_snyder_edd"] = edd
            del tmax
        if world == 1 and not small and extras and dtype == "float32":
            res.update(boundary_legs(Xs, lat, lon, df, plan, T, Gs, Rr))
        plan.close()
        return res

    cpu_dense_memo = {}
    c5_tables = {}                       # block-local? -> (rowptr, col, val) host CSR of the synthetic c5 table

    def run_dense_family(wl, plan=None, steps=None, warmup=None, keep_plan=False, share=False):
        """c2-dense / c4 (full matrix), c5-uniform[-f64] (entry lists), c5-block[-f64] (tile-sparse)."""
        steps, warmup = steps or a.steps, a.warmup if warmup is None else warmup
        T, rows_all, scal = rows_for(wl, share)
        T_job = sum(rows_all)
        f64 = wl.endswith("-f64")
        dt_name = "float64" if f64 else "float32"
        X = engine.synth_field(T, G, seed=1000 + rank, base=280.0, amp=60.0, dtype=dt_name)
        build = None
        if wl.startswith("c5"):
            # configs[4] "sparse CSR weights": the table is HANDED IN as host CSR arrays (3 GB; here written by the
            # library's generator of the synthetic tables, outside the timed build) and the plan is what
            # wagg_dense_create_from_csr* makes of it -- bit-equal to the plan generated on the device from the same
            # hashes (tests/test_gpu_round4.py)
            bl = wl.startswith("c5-block")
            fill = 0.952 if bl else 0.01
            if plan is None:
                if bl not in c5_tables:
                    c5_tables.clear()                            # one table (3 GB of host memory) at a time
                    c5_tables[bl] = engine.synth_table_csr(G, R, 2, fill, blocklocal=bl)
                rowptr, col, val = c5_tables[bl]
                t0 = time.perf_counter()
                plan = engine.DensePlan.from_csr(rowptr, col, val, G, R, dtype=dt_name)
                build = {"plan_build_s": round(time.perf_counter() - t0, 4), "library_s": round(plan.info["build_s"], 4),
                         "upload_s": round(plan.info["build_upload_s"], 4), "table_entries": int(len(col)),
                         "table_bytes": int(rowptr.nbytes + col.nbytes + val.nbytes),
                         "what": "wagg_dense_create_from_csr%s on host CSR arrays: upload, device radix sort + coalesce + "
                                 "denominators + form choice + packing" % ("_f64" if f64 else "")}
        else:
            fill, bl = 1.0, False
            plan = plan or engine.DensePlan.synth(G, R, seed=2)
        st = stepper(lambda out: plan.apply(X, out=out, ksplit=a.ksplit), T, rows_all, R, X.dtype)
        engine.profile_enable(True)       # event records only (no sync)
        dt, per_step = timed_steps(torch, dist, st.step, st.finish, steps, warmup, world)
        kms = engine.profile_read()
        engine.profile_enable(False)
        gok = gather_check(st, T)
        form = int(plan.info["form"])
        # one apply = one dominant-kernel launch (a full-form apply of several row blocks is one launch too)
        kavg = kernel_avg_ms(kms, warmup) * 1e-3
        if wl.startswith("c5-uniform"):
            nnz = int(plan.info["nnz"])
        elif wl.startswith("c5-block"):
            nnz = int(round(plan.info["n_tiles"] * (16 if f64 else 32) * 256 * fill))
        else:
            nnz = G * R
        flops = 2.0 * T * nnz                                   # algorithmic: 2 T nnz (dense: nnz = G R)
        traffic, tsrc = load_traffic(wl)
        kname = {0: "dense_mfma_kernel", 1: "dense_mfma_kernel<tiled>", 2: "spmm_kernel (vector ALU, entry lists)"}[form]
        if f64:
            kname = kname.replace("dense_mfma_kernel", "dense_mfma_kernel<double>").replace("spmm_kernel", "spmm_kernel<double>")
        # entry lists run on the vector ALU: fp32 FMA peak = the fp32 MFMA peak (157.3), fp64 FMA = 78.6
        peak = PEAK_F64_MFMA_TFLOPS if f64 else PEAK_F32_MFMA_TFLOPS
        res = {"workload": wl, "dtype": "f64" if f64 else "f32", "T": T, "T_job": T_job, "G": G, "R": R, "nnz": nnz,
               "value": T_job * G * R * steps / dt, "unit": "gridcell-region-timesteps/s", "steps": steps, "warmup": warmup,
               "ms_per_step": dt / steps * 1e3, **step_stats(per_step),
               "plan": {k: (int(v) if isinstance(v, int) else round(v, 4)) for k, v in plan.info.items()}, "scaling": scal,
               "roofline": {"bound": "mfma" if form != 2 else "valu", "achieved": flops / kavg / 1e12, "peak": peak,
                            "unit": "TFLOP/s", "frac": flops / kavg / 1e12 / peak,
                            "traffic": traffic, "traffic_source": tsrc, "kernel": kname,
                            "kernel_ms_avg": kavg * 1e3, "algorithmic_flops_per_launch": flops}}
        if gok is not None:
            res["gather_ok"] = gok
        if build:
            res["plan_build"] = build
            res["plan_build_s"] = build["plan_build_s"]
        if wl != "c2-dense":
            res["rows"] = ("one rank's share: %d of the job's %d rows (shard 0 of %d)"
                           % (T, C4_TOTAL if wl == "c4" else C5_TOTAL, a.shards or 8)) if world == 1 else "rows %r" % (rows_all,)
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            key = (fill, bl)                                    # c4 contracts the c2-dense operand: one CPU window serves both
            if key not in cpu_dense_memo:
                cpu_dense_memo[key] = cpu_baseline_dense(min(T, 365), G, R, 2, fill, bl)
            res["cpu_baseline"] = cpu_dense_memo[key]
        del X
        if not keep_plan:
            plan.close()
            plan = None
        return res, plan

    secondary = []
    sec_steps = max(1, min(a.steps, 5))
    if a.workload in DENSE_FAMILY:
        want_secondary = a.workload == "c2-dense" and world == 1 and not a.no_secondary
        main_res, plan = run_dense_family(a.workload, keep_plan=want_secondary or (a.workload == "c2-dense" and world > 1 and not a.no_secondary))
        scaling = main_res.get("scaling", "weak")
        if want_secondary:
            # every other BASELINE config on this GPU (N = 1): c4's rank share on the SAME 101 GB operand, then the rest
            r4, _ = run_dense_family("c4", plan=plan, steps=min(sec_steps, 3), warmup=1, share=True)
            secondary.append(r4)
            torch.cuda.empty_cache()
            # the segment-table steps take 0.04-0.4 ms: 100 of them after 10 warm-up (five would time 2 ms of a cold start)
            seg_steps = max(sec_steps, 100)
            secondary.append(run_sparse("float32", steps=seg_steps, warmup=10))        # c2-real: segment-table form, fp32, area weights
            secondary.append(run_sparse("float64", steps=seg_steps, warmup=10))        # c3-real: fp64 data, pop weights with backup fill
            secondary.append(run_sparse("float64", small=True, steps=seg_steps, warmup=10))   # c1
            for wl in ("c5-block", "c5-block-f64", "c5-uniform", "c5-uniform-f64"):
                torch.cuda.empty_cache()
                secondary.append(run_dense_family(wl, steps=sec_steps, warmup=2, share=True)[0])
        elif a.workload == "c2-dense" and world > 1 and not a.no_secondary:
            # multi-GPU run: after the weak-scaled headline the STRONG splits of the two 8-GPU configs BASELINE names --
            # configs[3] (10,950 rows over the ranks, on the operand already resident) and configs[4] (18,250 rows,
            # entry lists) -- so that the scaling line speaks to them too.  Every rank takes part (the gather is inside).
            a_shards = a.shards
            a.shards = world
            r4, _ = run_dense_family("c4", plan=plan, steps=2, warmup=1)
            torch.cuda.empty_cache()
            r5, _ = run_dense_family("c5-uniform", steps=3, warmup=1)
            a.shards = a_shards
            secondary.extend([r4, r5])
    elif a.workload == "c1":
        main_res = run_sparse("float64", small=True)
    else:
        main_res = run_sparse("float32" if a.workload == "c2-real" else "float64")
    T_job = main_res["T_job"]

    if rank == 0:
        wl = main_res["workload"]
        what = {"c1": "2-degree grid, 100 block regions, area-weighted, f64 (the reference's CPU-sized case)",
                "c2-dense": "area-weighted dense (gridcell x region) W, f32",
                "c4": "30-year daily series, dense W, f32", "c2-real": "area-weighted segment table, f32",
                "c3-real": "pop-weighted segment table with backup fill, f64",
                "c5-uniform": "ensemble x time rows, ~1 % of G x R non-zero at uniformly random positions, f32",
                "c5-block": "ensemble x time rows, ~1 % of G x R non-zero, block-local, f32",
                "c5-block-f64": "ensemble x time rows, ~1 % of G x R non-zero, block-local, f64",
                "c5-uniform-f64": "ensemble x time rows, ~1 % of G x R non-zero at uniformly random positions, f64"}[wl]
        line = {
            "metric": "gridcell-region-timesteps/sec", "value": main_res["value"],
            "unit": "gridcell-region-timesteps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": main_res["ms_per_step"], "median_ms": main_res.get("median_ms"), "min_ms": main_res.get("min_ms"),
            "higher_is_better": True, "scaling": scaling,
            **({"rehearsal": "WAGG_BENCH_REHEARSE=gloo: every rank on device 0, blocks gathered through host memory -- control flow "
                             "only, NOT a measurement"} if rehearse else {}),
            "scaling_measured": world > 1 and not rehearse,     # a one-GPU line says nothing about scaling; the driver computes efficiency from its N = 1, 2, 4, 8 runs
            "vs_baseline": None, "dtype": main_res["dtype"], "data": "synthetic",
            "config": {"workload": "%s: daily tas, T=%d rows on this GPU (%d in the job), grid G=%d, R=%d regions, %s; "
                                   "time axis sharded over %d GPU(s) + RCCL gather"
                                   % (wl, main_res["T"], T_job, main_res["G"], main_res["R"], what, world),
                       "T_per_gpu": main_res["T"], "T_job": T_job, "G": main_res["G"], "R": main_res["R"],
                       "parallelism": "time-shard x%d" % world},
            "roofline": main_res["roofline"],
            "cpu_baseline": main_res.get("cpu_baseline"),
        }
        for k in ("plan_build_s", "plan_build", "gather_ok", "host_resident", "dropin_ms", "agglev_ISO", "layout_gridcell_time",
                  "fused_tas_poly_1to4", "fused_snyder_edd"):
            if k in main_res:
                line[k] = main_res[k]
        if "nnz" in main_res:
            line["config"]["nnz"] = main_res["nnz"]
            line["config"]["plan"] = main_res["plan"]
        if secondary:
            line["secondary"] = secondary
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
