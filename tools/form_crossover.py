#!/usr/bin/env python3
"""Where do the device forms of a caller's weight table cross over?  (GPU box.)

The reference has ONE weights type (aggregations.py:64-73); here a table becomes one of four device forms.  This tool
builds the SAME caller table in each dense-family form by force (``flags = WAGG_DENSE_FORCE_*`` of
wagg_dense_create_from_csr) and times one rank shard of rows through each, next to what the library picks by itself and
to the per-row estimates its choice went by (wagg_dense_info.est_row_s).  Three sweeps:

  uniform     fill f of all (cell, region) pairs non-zero at uniformly random positions (c5 "uniform" at f = 1 %)
  blocklocal  every run of 64 cells touches a share p of the 256-region column tiles, filled to 30 % inside
  scatter     the c2-real segment table with a share q of its rows given a random region: segment-table form against the
              dense family (the drop-in's DENSE_SWITCH on gathered cells per timestep / grid cells)

Grid: --nlat x --nlon cells, --R regions, --T rows.  Prints one JSON line per point and a final table; the text that
profiles/r05_form_crossover.txt holds is this tool's output.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from climate_toolbox_amd import _lib, engine, synth
from climate_toolbox_amd.engine import DensePlan, SparsePlan

FORMS = ("full", "tiles", "entries")
FORM_NAME = {0: "full", 1: "tiles", 2: "entries"}


def time_apply(plan, X, reps_min=5, fill_s=0.4, **kw):
    out = torch.empty((X.shape[0], plan.R), dtype=X.dtype, device="cuda")
    t0 = time.perf_counter()
    plan.apply(X, out=out, **kw)
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    n_warm = max(1, int(0.15 / max(one, 1e-4)))
    for _ in range(n_warm):
        plan.apply(X, out=out, **kw)
    torch.cuda.synchronize()
    reps = max(reps_min, min(200, int(fill_s / max(one, 1e-4))))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        plan.apply(X, out=out, **kw)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2], ms[0], out


def blocklocal_table(G, R, share, fill_in, seed):
    """CSR arrays of a table in which every run of 64 cells touches round(share * n_nt) of the n_nt column tiles (256 regions
    each; which ones: a hash of the run), each touched (cell, region) pair kept with probability fill_in.  Generated on the
    device with torch (a tool's input generator, not product code), returned as host arrays."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    n_nt = (R + 255) // 256
    k = max(1, int(round(share * n_nt)))
    rows, cols, vals = [], [], []
    counts = torch.zeros(G, dtype=torch.int64, device="cuda")
    BL = 64 * 64                                         # cells per generated block
    for g0 in range(0, G, BL):
        g1 = min(G, g0 + BL)
        runs = torch.arange(g0 // 64, (g1 + 63) // 64, device="cuda")
        # tiles of a run: k distinct tiles starting at a hashed offset, stride coprime to n_nt
        start = (runs * 97) % n_nt
        tiles = (start[:, None] + torch.arange(k, device="cuda")[None, :] * 1) % n_nt            # (runs, k)
        occ = torch.zeros((len(runs), n_nt), dtype=torch.bool, device="cuda")
        occ.scatter_(1, tiles, True)
        occ_cells = occ.repeat_interleave(64, dim=0)[: g1 - g0]                                  # (cells, n_nt)
        mask = occ_cells.repeat_interleave(256, dim=1)[:, :R]
        mask &= torch.rand((g1 - g0, R), device="cuda", generator=g) < fill_in
        nz = mask.nonzero()
        counts[g0:g1] = torch.bincount(nz[:, 0], minlength=g1 - g0)
        cols.append(nz[:, 1].to(torch.int32).cpu())
        vals.append(torch.rand(len(nz), device="cuda", generator=g, dtype=torch.float64).mul_(0.9).add_(0.1).cpu())
        del mask, nz, occ, occ_cells
    rowptr = torch.zeros(G + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts.cpu(), 0)
    return rowptr.numpy(), torch.cat(cols).numpy(), torch.cat(vals).numpy()


def sweep_table(name, param, rowptr, col, val, G, R, X, dtype, forms=FORMS, max_full_bytes=None):
    rec = {"sweep": name, "param": param, "G": G, "R": R, "T": int(X.shape[0]), "dtype": dtype, "entries_in": int(len(col))}
    ms = {}
    ref = None
    for form in ("auto",) + tuple(forms):
        eb = 8 if dtype == "float64" else 4
        if form == "full" and max_full_bytes is not None and eb * G * ((R + 255) // 256 * 256) > max_full_bytes:
            rec["full"] = "skipped (matrix above the tool's byte cap)"
            continue
        try:
            t0 = time.perf_counter()
            plan = DensePlan.from_csr(rowptr, col, val, G, R, dtype=dtype, form=None if form == "auto" else form)
            tb = time.perf_counter() - t0
        except _lib.WaggError as e:
            rec[form] = "failed: %s" % str(e)[:120]
            continue
        med, mn, out = time_apply(plan, X)
        got = out[:4].cpu().numpy()
        if ref is None:
            ref = got
        err = float(np.nanmax(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)))
        info = plan.info
        if form == "auto":
            rec["picked"] = FORM_NAME[info["form"]]
            rec["nnz"] = int(info["nnz"])
            rec["walked_entries"] = int(info["walked_entries"])
            rec["est_ms"] = {"full": info["est_full_s"] * X.shape[0] * 1e3, "tiles": info["est_tiles_s"] * X.shape[0] * 1e3,
                             "entries": info["est_entries_s"] * X.shape[0] * 1e3}
            rec["n_tiles_stored_if_tiled"] = None
        if form == "tiles":
            rec["tile_share"] = info["n_tiles"] / float(info["n_kt"] * info["n_nt"])
        ms[form] = med
        rec[form] = {"ms": round(med, 4), "min_ms": round(mn, 4), "build_s": round(tb, 3), "max_rel_diff_vs_auto": err}
        plan.close()
        del plan, out
        torch.cuda.empty_cache()
    forced = {f: ms[f] for f in forms if f in ms}
    if forced and "auto" in ms:
        best = min(forced, key=forced.get)
        rec["fastest_forced"] = best
        rec["auto_over_fastest"] = round(ms["auto"] / forced[best], 4)
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nlat", type=int, default=360)
    ap.add_argument("--nlon", type=int, default=720)
    ap.add_argument("--R", type=int, default=24378)
    ap.add_argument("--T", type=int, default=2282)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--sweeps", default="uniform,blocklocal,scatter")
    ap.add_argument("--fills", default="0.003,0.01,0.03,0.06,0.10,0.15,0.30")
    ap.add_argument("--shares", default="0.05,0.25,0.50,0.75")
    ap.add_argument("--scatters", default="0,0.1,1.0")
    ap.add_argument("--fill-in", type=float, default=0.30)
    a = ap.parse_args()
    G, R, T = a.nlat * a.nlon, a.R, a.T
    X = engine.synth_field(T, G, seed=1000, base=280.0, amp=60.0, dtype=a.dtype)
    recs = []
    sweeps = a.sweeps.split(",")
    if "uniform" in sweeps:
        for f in [float(x) for x in a.fills.split(",")]:
            rowptr, col, val = engine.synth_table_csr(G, R, 2, f, blocklocal=False)
            recs.append(sweep_table("uniform", f, rowptr, col, val, G, R, X, a.dtype))
            del rowptr, col, val
    if "blocklocal" in sweeps:
        for p in [float(x) for x in a.shares.split(",")]:
            rowptr, col, val = blocklocal_table(G, R, p, a.fill_in, seed=5)
            recs.append(sweep_table("blocklocal", p, rowptr, col, val, G, R, X, a.dtype))
            del rowptr, col, val
    if "scatter" in sweeps:
        # the real segment table, a share q of its rows re-labelled with a random region; then (q = 1) m rows per land cell,
        # each with a random region, for the full region count and for a small one (the entry lists' X stream is paid once per
        # block of 688 regions): segment-table plan vs dense family -- the drop-in's choice between the two families
        from climate_toolbox_amd import aggregations as A
        lat, lon, df = synth.realistic_segments(nlat=a.nlat, nlon=a.nlon, R=R, string_labels=False)
        cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
        rng = np.random.default_rng(11)
        T2 = 365
        X2 = X[:T2]
        land = np.unique(cell)
        points = [("q", float(x), len(uniq)) for x in a.scatters.split(",")]
        points += [("m", m, Rm) for Rm in (len(uniq), 600) for m in (1, 2, 4, 8, 16, 32)]
        for kind, v, Rr in points:
            if kind == "q":
                c1, c2, w2 = cell, code.copy(), w
                mk = rng.random(len(c2)) < v
                c2[mk] = rng.integers(0, Rr, int(mk.sum()))
            else:
                c1 = np.repeat(land, int(v)).astype(np.int32)
                c2 = rng.integers(0, Rr, len(c1)).astype(np.int32)
                w2 = rng.uniform(0.1, 1.0, len(c1))
            rec = {"sweep": "scatter", "kind": kind, "param": v, "G": G, "R": Rr, "T": T2, "rows": int(len(c1))}
            sp = SparsePlan(c1, c2, w2, G, Rr, row_len=len(lon))
            med, mn, out = time_apply(sp, X2)
            ref = out[:4].cpu().numpy()
            rec["n_ucells_over_G"] = round(sp.info["n_ucells"] / G, 3)
            rec["n_giant"] = int(sp.info["n_giant"])
            rec["segment_table"] = {"ms": round(med, 4), "min_ms": round(mn, 4)}
            rec["dropin_wants_dense"] = bool(A._wants_dense(sp.info["n_ucells"], G, "TG", R=Rr, nseg=len(c1), is_f32=a.dtype == "float32"))
            sp.close()
            for form in ("auto", "tiles", "entries"):
                dp = DensePlan.from_segments(c1, c2, w2, G, Rr, dtype=a.dtype, form=None if form == "auto" else form)
                med, mn, out = time_apply(dp, X2)
                got = out[:4].cpu().numpy()
                rec["dense_" + form] = {"ms": round(med, 4), "min_ms": round(mn, 4), "form": FORM_NAME[dp.info["form"]],
                                        "max_rel_diff_vs_segment_table": float(np.nanmax(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)))}
                dp.close()
                del dp
                torch.cuda.empty_cache()
            best_dense = min(rec["dense_tiles"]["ms"], rec["dense_entries"]["ms"])
            rec["dense_over_segment_table"] = round(best_dense / rec["segment_table"]["ms"], 3)
            print(json.dumps(rec), flush=True)
            recs.append(rec)
    # ---- table
    print("\n== form crossover, G=%d R=%d T=%d %s ==" % (G, R, T, a.dtype))
    print("%-11s %8s %12s %10s | %9s %9s %9s | %9s | %-8s %-8s %s" % ("sweep", "param", "nnz", "walked/nnz", "full ms", "tiles ms", "entr. ms",
                                                                       "auto ms", "picked", "fastest", "auto/fastest"))
    for r in recs:
        if r["sweep"] == "scatter":
            continue
        g = lambda k: ("%9.3f" % r[k]["ms"]) if isinstance(r.get(k), dict) else "%9s" % "-"
        print("%-11s %8.4g %12d %10.2f | %s %s %s | %s | %-8s %-8s %s" % (
            r["sweep"], r["param"], r.get("nnz", 0), r.get("walked_entries", 0) / max(1, r.get("nnz", 1)), g("full"), g("tiles"), g("entries"),
            g("auto"), r.get("picked"), r.get("fastest_forced"), r.get("auto_over_fastest")))
    for r in recs:
        if r["sweep"] != "scatter":
            continue
        chosen, other = ((r["dense_auto"]["ms"], r["segment_table"]["ms"]) if r["dropin_wants_dense"] else (r["segment_table"]["ms"], r["dense_auto"]["ms"]))
        right = chosen <= 1.25 * other                      # (the family the drop-in takes is at most 25 % behind the other)
        print("scatter %s=%-5g R=%-6d rows=%-9d n_ucells/G=%-8.3f giants=%-6d segment-table %9.3f ms | dense tiles %9.3f  entries %9.3f  auto(%s) %9.3f | "
              "wants_dense=%-5s %s" % (r["kind"], r["param"], r["R"], r["rows"], r["n_ucells_over_G"], r["n_giant"], r["segment_table"]["ms"],
                                       r["dense_tiles"]["ms"], r["dense_entries"]["ms"], r["dense_auto"]["form"], r["dense_auto"]["ms"],
                                       r["dropin_wants_dense"], "ok" if right else "MORE THAN 25 % BEHIND"))


if __name__ == "__main__":
    main()
