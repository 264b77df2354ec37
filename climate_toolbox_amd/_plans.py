"""The plan side of the drop-in: which FAMILY serves a table (segment-table gather or the dense family) and why, the leased
plan cache with its count / byte budget, and the replicas of a plan for several row-block pipelines.

Split out of aggregations.py in round 6; aggregations.py re-exports every name (``HOST_DEVICES`` stays the public knob THERE
and is handed in; the cache limits are set HERE: ``_plans._PLAN_CACHE_MAX``)."""
from __future__ import annotations

import os
import threading
from collections import OrderedDict

from ._lib import FORM_ENTRIES, WaggError
from ._memo import _CACHE_LOCK, _fingerprint
from .engine import DensePlan, SparsePlan

_PLAN_CACHE: "OrderedDict[str, SparsePlan]" = OrderedDict()
_PLAN_CACHE_MAX = 8                 # plans
_PLAN_CACHE_MAX_FRAC = 0.5          # ... and at most this share of the device's memory (dense plans are GBs)
_REPLICA_MAX_BYTES = 8 << 30          # plans above this (the 101 GB dense operand) are not replicated


def _host_devices(setting):
    """Device ordinals of the row-block pipelines for ``aggregations.HOST_DEVICES = setting`` ([] = the current device only)."""
    import torch
    HOST_DEVICES = setting
    if HOST_DEVICES is None:
        return []
    if isinstance(HOST_DEVICES, str):
        if HOST_DEVICES != "all":
            raise ValueError('HOST_DEVICES must be None, "all" or a list of device ordinals')
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            return []                  # one process per GPU: the other devices belong to the other ranks
        return list(range(torch.cuda.device_count()))
    return [int(d) for d in HOST_DEVICES]


def _host_quantum(plan):
    """Rows one launch of the plan's kernel handles well (what wagg_host_block_plan sizes the row blocks by)."""
    if isinstance(plan, DensePlan):
        f64 = plan.dtype == "float64"
        if plan.info["form"] == FORM_ENTRIES:
            return 64 if f64 else 128
        return 176 if f64 else 368
    return 64


def _host_replicas(plan, n_rows, row_bytes, setting=None):
    """Replicas of a leased plan for the other pipelines of HOST_DEVICES (= ``setting``), built once and kept with the plan; () when one
    device serves the call (the default, a field of few blocks, a plan too large to copy around).  The plan itself
    serves the first pipeline on its own device (the first pipeline at all when its device is not listed, in place of
    that entry): len(devices) pipelines in every case."""
    devs = _host_devices(setting)
    if len(devs) < 2 or _plan_bytes(plan) > _REPLICA_MAX_BYTES:
        return ()
    from ._lib import host_block_plan
    if host_block_plan(n_rows, row_bytes, _host_quantum(plan), len(devs))[1] < 2 * len(devs):
        return ()                      # not enough blocks for every device to overlap its copies with its kernels
    own = devs.index(plan.device) if plan.device in devs else 0
    cache = plan.__dict__.setdefault("_replicas", {})
    out = []
    for slot, d in enumerate(devs):
        if slot == own:
            continue                   # the plan itself
        key = (slot, d)
        if key not in cache:
            cache[key] = plan.replica(d)
        out.append(cache[key])
    return tuple(out)


# Which FAMILY serves a table -- the segment-table gather or the dense family -- by estimated time per row of X, each at its
# measured rate (tools/form_crossover.py, profiles/r05_form_crossover.txt, docs/HISTORY.md (b)):
#   segment table   n_ucells cell slots gathered per row at _SPARSE_CELLS_PER_S (2.7e11 / s in fp32: 0.13 ms for 0.46 G
#                   compact cells x 365 rows, 0.64 ms for 1.8 G scattered ones; fp64 moves twice the bytes per cell)
#   dense family    at best its entry lists: never faster than the X stream that each block of 688 regions pulls through the
#                   LDS-DMA path (b G n_rb bytes per row at 10.1e12 B/s), nor than 2 x 1.3 nseg flop at the entry-loop rate
# The dense family takes over when the gather would cost _DENSE_MARGIN times as much (the estimate of the dense side is its
# floor; building it costs more and holds more memory).  Round 4's rule -- n_ucells > 16 G whatever the region count -- sat
# on the wrong side for small region counts (R = 600: the dense family wins from n_ucells ~ 0.2 G).
DENSE_SWITCH = 16.0   # the old rule, kept for callers that do not know R (then: n_ucells > DENSE_SWITCH * G)
_SPARSE_CELLS_PER_S = {True: 2.7e11, False: 1.8e11}      # is_f32 -> gathered cell slots per second
_DENSE_MARGIN = 1.5
_DENSE_BUILD_BYTES_PER_ROW = 64      # device scratch of wagg_dense_create_from_segments while it builds (48 + 16 per row)
_ENOMEM = -3                         # wagg.h WAGG_ENOMEM


def _dense_floor_s(G, R, nseg, is_f32):
    """Seconds per row of X the dense family cannot beat for this table (its entry-list form)."""
    b = 4 if is_f32 else 8
    n_rb = -(-int(R) // (16 * 43))
    loop = 2.0 * 1.3 * float(nseg) / (30e12 if is_f32 else 13e12)
    pack = 2.0 * b * float(G) / 5.3e12                  # X is packed once per apply (read + write at the HBM rate)
    return max(b * float(G) * n_rb / 10.1e12, loop) + pack


def _wants_dense(n_ucells, G, layout, R=None, nseg=None, is_f32=True):
    """Device-family choice for one weights table.  The gather form fetches ``n_ucells`` cell slots
    per timestep; when regions are scattered all over the grid (e.g. <=1 % non-zeros at random
    columns: every region is a multi-chunk "giant") that is many times the grid itself and the
    dense-family forms (MFMA contraction of the stored tiles in fp32 or fp64, entry lists for very
    sparse tables), whose cost does not depend on where a region's cells lie, are faster -- for a
    (time, gridcell) problem."""
    if layout != "TG":
        return False
    if R is None or nseg is None:
        return n_ucells > DENSE_SWITCH * G
    return n_ucells / _SPARSE_CELLS_PER_S[bool(is_f32)] > _DENSE_MARGIN * _dense_floor_s(G, R, nseg, is_f32)


def _dense_bytes(G, R, is_f32=True, nseg=None):
    """Upper bound of what a dense-family plan of this table holds in HBM: the full matrix, or -- knowing the number of rows --
    what the library's form choice can store for that many pairs: entry lists (8 B per pair in fp32, 16 B in fp64), or the
    tile-sparse form, which it only takes while the stored tiles' flops beat the entry loop, i.e. up to ~1.1e-3 (fp32) /
    2.5e-3 (fp64) tiles of 32 KB per walked entry (csrc/wagg_dense.hip: table_form_cost) -- 64 / 160 bytes per row."""
    eb = 4 if is_f32 else 8
    full = eb * ((int(G) + 31) // 32 * 32) * ((int(R) + 255) // 256 * 256)
    if nseg is not None:
        # (+ the entry lists' padding: every (region block, chunk, wave) bucket is filled up to a multiple of 8 entries -- at most
        #  7 x 8 B per bucket, 16 waves per (block of 688 regions, chunk of 128 cells): what dominates for small scattered tables)
        n_buckets = 16 * (-(-int(R) // 688)) * (-(-int(G) // 128))
        return min(full, (64 if is_f32 else 160) * int(nseg) + 7 * (8 if is_f32 else 16) * n_buckets)
    return full


def _prefer_dense(n_ucells, G, R, is_f32, layout, free_bytes, nseg=None):
    """The family choice as one predicate (tests, tools): the gather would cost more AND the dense-family plan fits."""
    return (_wants_dense(n_ucells, G, layout, R=R, nseg=nseg if nseg is not None else n_ucells, is_f32=is_f32)
            and _dense_bytes(G, R, is_f32, nseg=nseg) < 0.6 * free_bytes)


def _plan_bytes(plan):
    return int(plan.info.get("w_bytes", 0)) if isinstance(plan, DensePlan) else 16 * int(plan.info.get("nnz", 0))


def _evict_plans(byte_budget, keep):
    """Close cached plans, oldest first, until at most ``keep`` remain and they hold no more than
    ``byte_budget`` bytes of device memory; returns the bytes released.  Plans leased to a running call
    (their lock is held) are passed over.  Caller holds _CACHE_LOCK."""
    freed = 0
    held = sum(_plan_bytes(p) for p in _PLAN_CACHE.values())
    for key in list(_PLAN_CACHE):
        if not (len(_PLAN_CACHE) > keep or held > max(byte_budget, 0)):
            break
        old = _PLAN_CACHE[key]
        if not old._lease.acquire(blocking=False):
            continue                                   # in use by another thread: not ours to close
        try:
            del _PLAN_CACHE[key]
            b = _plan_bytes(old)
            old.close()
        finally:
            old._lease.release()
        held -= b
        freed += b
    return freed


def _drop_plan(plan):
    """Forget a plan that failed on the device (a poisoned plan keeps failing: its timeout word is sticky)."""
    with _CACHE_LOCK:
        for key, p in list(_PLAN_CACHE.items()):
            if p is plan:
                del _PLAN_CACHE[key]
    plan.close()


_BUILDING = {}        # plan key -> threading.Event of the thread that is building that plan right now (under _CACHE_LOCK)


def _plan_for(cell_idx, codes, w_eff, G, R, row_len, is_f32=False, layout="TG", prepared=None):
    """The cached plan of this table, LEASED: ``plan._lease`` is held on return and the caller releases
    it when its device work is done (see _CACHE_LOCK).  A table that is not cached is built by ONE thread, outside the
    cache lock (a dense-family plan can be GBs: other tables must not wait for it); threads that want the same table
    meanwhile wait for that build instead of starting their own."""
    if prepared is not None:
        key = prepared.plan_key(cell_idx, G, R, row_len, is_f32, layout)       # hashed once per (table, grid)
    else:
        key, _ = _fingerprint(cell_idx, codes, w_eff, extra=repr((int(G), int(R), int(row_len), bool(is_f32), layout)))
    while True:
        with _CACHE_LOCK:
            plan = _PLAN_CACHE.get(key)
            pending = None
            if plan is not None:
                _PLAN_CACHE.move_to_end(key)
            else:
                pending = _BUILDING.get(key)
                if pending is None:
                    _BUILDING[key] = threading.Event()
        if plan is not None:
            plan._lease.acquire()                  # waits for a thread that is applying the same table
            if plan._h.value:
                return plan
            plan._lease.release()                  # closed in the meantime (failed on the device): forget it, build anew
            with _CACHE_LOCK:
                if _PLAN_CACHE.get(key) is plan:
                    del _PLAN_CACHE[key]
            continue
        if pending is not None:
            pending.wait()                         # another thread builds this very table: take its plan from the cache
            continue
        break
    import torch
    plan = None
    try:
        with _CACHE_LOCK:
            free_bytes, total_bytes = torch.cuda.mem_get_info()
            # what the library's scratch pool keeps between builds (up to 1/16 of the device, the ~11 GB arena of a c5 build
            # among it) shows as used memory but is given back on demand: an allocation of the library that runs short frees
            # it and tries again (wagg_scratch.hip), so it counts as free here
            from . import _lib as _libmod
            if _libmod._lib is not None:
                free_bytes += int(_libmod._lib.wagg_scratch_bytes())
            # keep the cache under its plan count and byte budget (what a sparse plan adds is a few MB)
            free_bytes += _evict_plans(_PLAN_CACHE_MAX_FRAC * total_bytes, keep=_PLAN_CACHE_MAX - 1)
        # the finished plan plus what the device-side build holds while it runs (csrc/wagg_build.h: 48 bytes of arena per
        # table row -- the sort's key / value pairs and the uploaded table -- and 16 per distinct pair)
        need = _dense_bytes(G, R, is_f32, nseg=len(cell_idx)) + _DENSE_BUILD_BYTES_PER_ROW * len(cell_idx)

        def dense_fits():
            # only now is the dense byte budget charged: cached plans can be given back (oldest first) before
            # the dense form is declined for lack of memory
            nonlocal free_bytes
            if need >= 0.6 * free_bytes:
                with _CACHE_LOCK:
                    free_bytes += _evict_plans(_PLAN_CACHE_MAX_FRAC * total_bytes - need, keep=_PLAN_CACHE_MAX - 1)
            return need < 0.6 * free_bytes

        dt = "float32" if is_f32 else "float64"
        # a table with far more rows than grid cells (c5: ~244 per cell) cannot win in the gather
        # form: go to the dense-family form directly instead of building the sparse plan first just to
        # read its statistics

        def dense_plan():
            # a build that runs out of device memory after all (another process took it meanwhile; the estimate was short)
            # is not the caller's problem: cached plans go and it is tried once more, then the segment-table form serves
            if len(cell_idx) >= 2 ** 31:           # wagg_dense_create_from_segments: at most 2^31 - 1 rows
                return None
            for attempt in (0, 1):
                try:
                    return DensePlan.from_segments(cell_idx, codes, w_eff, G, R, dtype=dt)
                except WaggError as e:
                    if getattr(e, "code", None) != _ENOMEM:
                        raise
                    if attempt == 0:
                        with _CACHE_LOCK:
                            _evict_plans(0, keep=0)
            return None

        # (n_ucells >= the table's distinct cells: a table with that many rows per cell needs no sparse plan to know)
        sure = len(cell_idx) > 4 * DENSE_SWITCH * G and _wants_dense(len(cell_idx) / 4.0, G, layout, R=R, nseg=len(cell_idx), is_f32=is_f32)
        if sure and dense_fits():
            plan = dense_plan()
        if plan is None:
            plan = SparsePlan(cell_idx, codes, w_eff, G, R, row_len=row_len)
            if not sure and _wants_dense(plan.info["n_ucells"], G, layout, R=R, nseg=len(cell_idx), is_f32=is_f32) and dense_fits():
                dense = dense_plan()
                if dense is not None:
                    plan.close()
                    plan = dense
        plan._lease.acquire()
    finally:
        with _CACHE_LOCK:
            if plan is not None and plan._h.value:
                _PLAN_CACHE[key] = plan
            _BUILDING.pop(key).set()               # (a failed build wakes the waiters too: the first of them tries again)
    return plan


def _clear_plans():
    """Close every cached plan that no call is using (clear_caches)."""
    with _CACHE_LOCK:
        for key in list(_PLAN_CACHE):
            plan = _PLAN_CACHE[key]
            if plan._lease.acquire(blocking=False):
                try:
                    del _PLAN_CACHE[key]
                    plan.close()
                finally:
                    plan._lease.release()
