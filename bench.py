#!/usr/bin/env python3
"""bench.py -- gridcell-region-timesteps/s of the weighted grid->region aggregation on MI355X.

A "step" is one pass of the hot path over one batch: out[T, R] = (X[T, G] . W[G, R]) / den[R].

Default workload (N=1) = BASELINE.json configs[1] in its dense north-star form ("c2-dense"):
T=365 daily steps, 0.25-degree grid G=720x1440=1,036,800, R=24,378 impact regions, fp32, W a
dense (gridcell x region) matrix generated on the device (101 GB, hash of (g, r, seed)); inputs
are resident in HBM before the timed region.  The same run also times "c2-real" (the sparse
segment-table form of the same config: ~4e5 segments, HBM-bound) and reports it under
"secondary".  With --gpus N (launched by torch.distributed.run, one rank per GPU) the time axis is
sharded: every rank aggregates its own 365-row shard (weak scaling) and the region time series
are reassembled on rank 0 with an RCCL gather inside the timed region.

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
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2-dense", choices=["c2-dense", "c2-real", "c3-real", "c5-block", "c5-uniform"])
    ap.add_argument("--T", type=int, default=365)
    ap.add_argument("--nlat", type=int, default=720)
    ap.add_argument("--nlon", type=int, default=1440)
    ap.add_argument("--R", type=int, default=24378)
    ap.add_argument("--ksplit", type=int, default=0)
    ap.add_argument("--land-frac", type=float, default=0.30, help="experiment knob for c2-real")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def load_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def sparse_algorithmic_bytes(T, G, R, nnz, b):
    # b*T*G + (b+4)*nnz + 4*(G+1) + b*T*R + b*R
    return b * T * G + (b + 4) * nnz + 4 * (G + 1) + b * T * R + b * R


def timed_steps(torch, dist, step, steps, warmup, world):
    world = world if not (dist.is_available() and dist.is_initialized()) else max(world, 2)   # forced-dist rehearsal
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_baseline_dense(T, G, R, seed_w):
    """The oracle's dense restatement (oracle/wagg_oracle.c, kind "port") on a bounded window of
    the same workload: all T rows x Gw cells x Rw regions, threads = OpenMP default."""
    import numpy as np
    from oracle import c_oracle
    Gw, Rw = min(G, 32768), min(R, 8192)
    rng = np.random.default_rng(0)
    X = (280.0 + 30.0 * rng.standard_normal((T, Gw))).astype(np.float32)
    c_oracle.dense_synth(X[:8], 0, min(Gw, 256), R, 0, min(Rw, 256), seed_w)     # warm
    t0 = time.perf_counter()
    c_oracle.dense_synth(X, 0, Gw, R, 0, Rw, seed_w)
    dt = time.perf_counter() - t0
    return {"value": T * Gw * Rw / dt, "unit": "gridcell-region-timesteps/s", "cores": c_oracle.threads(),
            "kind": "port", "wall_s": round(dt, 3),
            "sample": "oracle/wagg_oracle.c dense (fp64 accumulate, OpenMP) on T=%d x %d cells x %d regions "
                      "of the c2-dense operands (window of the 1,036,800 x 24,378 problem)" % (T, Gw, Rw)}


def cpu_baseline_sparse(X_host, cell, codes, w_eff, R, G, T_sample):
    """The oracle's faithful single-threaded restatement (gather -> fp64 multiply -> group-sum ->
    divide, like the reference) on the first T_sample timesteps of the c2-real workload."""
    import numpy as np
    from oracle import c_oracle
    c_oracle.segments(X_host[:1], cell, codes, w_eff, R)
    t0 = time.perf_counter()
    c_oracle.segments(X_host[:T_sample], cell, codes, w_eff, R)
    dt = time.perf_counter() - t0
    return {"value": T_sample * G * R / dt, "unit": "gridcell-region-timesteps/s", "cores": 1, "kind": "port",
            "wall_s": round(dt, 3), "nnz_timesteps_per_s": T_sample * len(cell) / dt,
            "sample": "oracle/wagg_oracle.c segments_%s, first %d of the timesteps, full segment table"
                      % ("f32" if X_host.dtype == np.float32 else "f64", T_sample)}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (a.gpus, a.gpus))
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
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from climate_toolbox_amd import engine, synth
    from climate_toolbox_amd.timeshard import gather_time_shards

    T, G, R = a.T, a.nlat * a.nlon, a.R
    # every rank owns its own T-row shard of the global (T*world) x G field (weak scaling)
    X = engine.synth_field(T, G, seed=1000 + rank, base=280.0, amp=60.0, dtype="float32")
    result = {}

    def run_sparse(dtype):
        lat, lon, df = synth.realistic_segments(nlat=a.nlat, nlon=a.nlon, R=R, seed=2, land_frac=a.land_frac)
        wname = "areawt" if dtype == "float32" else "popwt"
        cell, codes, w_eff, uniq = synth.code_segments(df, lat, lon, wname, "hierid")
        plan = engine.SparsePlan(cell, codes, w_eff, G, len(uniq), row_len=a.nlon)
        Xs = X if dtype == "float32" else X.double()
        out = torch.empty((T, len(uniq)), dtype=Xs.dtype, device="cuda")
        gathered = [torch.empty((T * world, len(uniq)), dtype=Xs.dtype, device="cuda") if use_dist and rank == 0 else None]

        def step():
            plan.apply(Xs, out=out)
            if use_dist:
                gather_time_shards(out, dst=0, out=gathered[0])

        engine.profile_enable(True)       # event records only, no synchronisation
        dt = timed_steps(torch, dist, step, a.steps, a.warmup, world)
        kms = engine.profile_read()[a.warmup:]
        engine.profile_enable(False)
        b = 4 if dtype == "float32" else 8
        nnz = plan.info["nnz"]
        Rr = len(uniq)
        kavg = sum(kms) / len(kms) * 1e-3
        abytes = sparse_algorithmic_bytes(T, G, Rr, nnz, b)
        wl = "c2-real" if dtype == "float32" else "c3-real"
        res = {
            "workload": wl, "dtype": "f32" if b == 4 else "f64", "T": T, "G": G, "R": Rr, "nnz": int(nnz),
            "value": T * world * G * Rr * a.steps / dt, "unit": "gridcell-region-timesteps/s",
            "ms_per_step": dt / a.steps * 1e3, "nnz_timesteps_per_s": T * world * nnz * a.steps / dt,
            "plan": {k: int(v) for k, v in plan.info.items()},
            "roofline": {"bound": "hbm", "achieved": abytes / kavg / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": abytes / kavg / 1e9 / PEAK_HBM_GBS, "traffic": load_traffic(wl),
                         "kernel": "sparse_lc_kernel" if b == 4 else "sparse_stream_kernel", "kernel_ms_avg": kavg * 1e3,
                         "algorithmic_bytes_per_launch": abytes},
        }
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_sparse(Xs[:64].cpu().numpy(), cell, codes, w_eff, Rr, G, min(T, 64))
        if dtype == "float32" and world == 1:
            # fused tas_poly: (tas - 273.15)^p, p = 1..4, one pass over the field
            pout = torch.empty((4, T, Rr), dtype=Xs.dtype, device="cuda")
            for _ in range(2):
                plan.apply_poly(Xs, -273.15, 4, out=pout)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                plan.apply_poly(Xs, -273.15, 4, out=pout)
            torch.cuda.synchronize()
            pdt = (time.perf_counter() - t0) / a.steps
            res["fused_tas_poly_1to4"] = {"ms_per_step": pdt * 1e3, "value": 4 * T * G * Rr / pdt,
                                          "unit": "gridcell-region-timesteps/s (4 powers)"}
        return res

    def run_dense():
        dense = engine.DensePlan.synth(G, R, seed=2)
        out = torch.empty((T, R), dtype=torch.float32, device="cuda")
        gathered = [torch.empty((T * world, R), dtype=torch.float32, device="cuda") if use_dist and rank == 0 else None]

        def step():
            dense.apply(X, out=out, ksplit=a.ksplit)
            if use_dist:
                gather_time_shards(out, dst=0, out=gathered[0])

        engine.profile_enable(True)       # event records only (no sync): negligible next to 0.1 s
        dt = timed_steps(torch, dist, step, a.steps, a.warmup, world)
        kms = engine.profile_read()[a.warmup:]
        engine.profile_enable(False)
        kavg = sum(kms) / len(kms) * 1e-3
        flops = 2.0 * T * G * R
        res = {
            "workload": "c2-dense", "dtype": "f32", "T": T, "G": G, "R": R,
            "value": T * world * G * R * a.steps / dt, "unit": "gridcell-region-timesteps/s",
            "ms_per_step": dt / a.steps * 1e3,
            "roofline": {"bound": "mfma", "achieved": flops / kavg / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": flops / kavg / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "traffic": load_traffic("c2-dense"), "kernel": "dense_mfma_kernel",
                         "kernel_ms_avg": kavg * 1e3, "algorithmic_flops_per_launch": flops},
        }
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_dense(T, G, R, 2)
        dense.close()
        return res

    def run_blocklocal(uniform=False):
        """c5, ~1 % of G x R non-zero, T defaults to one rank's share of 50 x 365 days.
        block-local: every 64-cell run touches 256 regions -> tile-sparse MFMA form.  uniform: the
        non-zeros sit at uniformly random positions -> no tile of W is empty, full dense form (the
        worst case: 100x the algorithmic flops are executed)."""
        Tb = a.T if a.T != 365 else 2282
        Xb = engine.synth_field(Tb, G, seed=1000 + rank, base=280.0, amp=60.0, dtype="float32")
        plan = engine.DensePlan.synth(G, R, seed=2, fill=0.01) if uniform else engine.DensePlan.synth_blocklocal(G, R, seed=2)
        out = torch.empty((Tb, R), dtype=torch.float32, device="cuda")

        def step():
            plan.apply(Xb, out=out)
            if use_dist:
                gather_time_shards(out, dst=0)

        engine.profile_enable(True)
        dt = timed_steps(torch, dist, step, a.steps, a.warmup, world)
        kms = engine.profile_read()[a.warmup:]
        engine.profile_enable(False)
        kavg = sum(kms) / len(kms) * 1e-3
        nnz = int(0.01 * G * R) if uniform else int(round(plan.info["n_tiles"] * 32 * 256 * 0.952))
        flops = 2.0 * Tb * nnz
        res = {"workload": "c5-uniform" if uniform else "c5-block", "dtype": "f32", "T": Tb, "G": G, "R": R, "nnz": nnz,
               "value": Tb * world * G * R * a.steps / dt, "unit": "gridcell-region-timesteps/s",
               "ms_per_step": dt / a.steps * 1e3, "plan": dict(plan.info),
               "roofline": {"bound": "mfma", "achieved": flops / kavg / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                            "unit": "TFLOP/s", "frac": flops / kavg / 1e12 / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                            "kernel": "dense_mfma_kernel" if uniform else "dense_mfma_kernel<tiled>", "kernel_ms_avg": kavg * 1e3,
                            "algorithmic_flops_per_launch": flops}}
        plan.close()
        return res

    if a.workload in ("c5-block", "c5-uniform"):
        main_res = run_blocklocal(uniform=a.workload == "c5-uniform")
        secondary = []
    elif a.workload == "c2-dense":
        main_res = run_dense()
        secondary = []
        if world == 1 and not a.no_secondary:
            torch.cuda.empty_cache()
            secondary.append(run_sparse("float32"))          # c2-real: segment-table form, fp32, area weights
            secondary.append(run_sparse("float64"))          # c3-real: fp64 data, pop weights with backup fill
    else:
        main_res = run_sparse("float32" if a.workload == "c2-real" else "float64")
        secondary = []

    if rank == 0:
        line = {
            "metric": "gridcell-region-timesteps/sec", "value": main_res["value"],
            "unit": "gridcell-region-timesteps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": main_res["dtype"], "data": "synthetic",
            "config": {"workload": "%s: daily tas T=%d per GPU, 0.25deg grid %dx%d (G=%d), R=%d regions, "
                                   "%s, time axis sharded over %d GPU(s) + RCCL gather"
                                   % (main_res["workload"], main_res["T"], a.nlat, a.nlon, G, main_res["R"],
                                      "%s weights (~1 %% of G x R non-zero), fp32" % main_res["workload"][3:] if main_res["workload"].startswith("c5-")
                                      else "area-weighted, %s" % main_res["dtype"], world),
                       "T_per_gpu": main_res["T"], "G": G, "R": main_res["R"], "parallelism": "time-shard x%d" % world},
            "roofline": main_res["roofline"],
            "cpu_baseline": main_res.get("cpu_baseline"),
        }
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
    main()
