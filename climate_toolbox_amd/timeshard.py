"""Time-axis sharding across the GPUs of one node.

Output row t depends only on input row t, so the leading (time / ensemble x time) axis is cut
into `world` contiguous blocks, one process per GPU, no exchange during compute; ONE collective
at the end reassembles the (T, R) region time series on a root rank (RCCL over xGMI when the
process group is "nccl"; the same code runs on "gloo" for the CPU tests).  Blocks land directly in
their rows of the destination tensor: equal blocks through ``dist.gather`` into views, ragged blocks
through grouped point-to-point transfers into views -- no padding, no concatenation.
"""
from __future__ import annotations


def shard_bounds(T, world):
    """[(start, stop)] per rank: ceil(T/world) rows for the first T mod world ranks."""
    base, extra = divmod(int(T), int(world))
    out, s = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((s, s + n))
        s += n
    return out


class _Done:
    def wait(self):
        return True


class _Works:
    def __init__(self, works):
        self._works = list(works)

    def wait(self):
        for w in self._works:
            w.wait()
        return True


def gather_time_shards(out_local, dst=0, rows=None, group=None, out=None, async_op=False):
    """Gather per-rank (T_rank, R) blocks on `dst`; returns the (sum T_rank, R) tensor there and
    None elsewhere.  `rows` = per-rank row counts when they differ.  `out` (a contiguous
    (sum rows, R) tensor on `dst`) receives the blocks in place; it is allocated when omitted.
    With ``async_op`` the call returns ``(result, handle)`` right after the transfers were queued;
    ``handle.wait()`` makes the result (and the reuse of ``out_local``) safe."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rows is None:
        rows = [out_local.shape[0]] * world
    rows = [int(n) for n in rows]
    if len(rows) != world or out_local.shape[0] != rows[rank]:
        raise ValueError("rank %d holds %d rows, rows = %r" % (rank, out_local.shape[0], rows))
    block = out_local if out_local.is_contiguous() else out_local.contiguous()
    total, tail = sum(rows), tuple(block.shape[1:])
    if rank == dst:
        if out is None:
            out = torch.empty((total,) + tail, dtype=block.dtype, device=block.device)
        elif tuple(out.shape) != (total,) + tail or not out.is_contiguous() or out.dtype != block.dtype:
            raise ValueError("out must be a contiguous %r %s tensor" % ((total,) + tail, block.dtype))
    starts = [sum(rows[:r]) for r in range(world)]
    handle = _Done()
    if all(n == rows[0] for n in rows):
        views = [out[s:s + n] for s, n in zip(starts, rows)] if rank == dst else None
        w = dist.gather(block, gather_list=views, dst=dst, group=group, async_op=async_op)
        if async_op and w is not None:
            handle = w
    else:
        # ragged shards: every block goes straight into its rows of `out` (grouped send/recv)
        ops = []
        if rank == dst:
            if rows[rank]:
                out[starts[rank]:starts[rank] + rows[rank]].copy_(block)
            for r in range(world):
                if r != dst and rows[r]:
                    ops.append(dist.P2POp(dist.irecv, out[starts[r]:starts[r] + rows[r]], _global(r, group), group))
        elif rows[rank]:
            ops.append(dist.P2POp(dist.isend, block, _global(dst, group), group))
        works = dist.batch_isend_irecv(ops) if ops else []
        if async_op:
            handle = _Works(works)
        else:
            _Works(works).wait()
    result = out if rank == dst else None
    return (result, handle) if async_op else result


def _global(rank_in_group, group):
    import torch.distributed as dist
    return rank_in_group if group is None else dist.get_global_rank(group, rank_in_group)


def aggregate_time_sharded(apply_fn, X_local, rows=None, dst=0, group=None):
    """apply_fn(X_local) -> (T_rank, R) on this rank's device, then the gather above."""
    return gather_time_shards(apply_fn(X_local), dst=dst, rows=rows, group=group)


class ShardedStep:
    """One benchmark / production step of a time-sharded aggregation with the reassembly of step k
    overlapped with the compute of step k + 1: results alternate between two local buffers, the
    gather of a buffer is queued asynchronously (RCCL runs it on its own stream) and waited for
    after the NEXT step's compute, just before that step's gather is queued (both write the same
    destination rows, so at most one gather is ever in flight), or in ``finish()``.

    apply_fn(out_buffer) must write this rank's (T_rank, R) block into ``out_buffer``."""

    def __init__(self, apply_fn, make_buffer, rows=None, dst=0, group=None, distributed=True):
        self.apply_fn, self.rows, self.dst, self.group = apply_fn, rows, dst, group
        self.distributed = distributed
        self.bufs = [make_buffer(), make_buffer()]
        self.pending = [None, None]
        self.gathered = None            # the destination tensor on `dst` (allocated by the first gather)
        self.k = 0

    def step(self):
        i = self.k & 1
        if self.pending[i] is not None:
            self.pending[i].wait()
            self.pending[i] = None
        self.apply_fn(self.bufs[i])
        if self.distributed:
            # the previous step's gather writes the same rows of `gathered`: it has had this step's compute to
            # finish in, and must be complete before the next one is queued (gloo completes asynchronous
            # collectives on several threads in no fixed order: a late gather k could overwrite gather k + 1)
            if self.pending[i ^ 1] is not None:
                self.pending[i ^ 1].wait()
                self.pending[i ^ 1] = None
            res, handle = gather_time_shards(self.bufs[i], dst=self.dst, rows=self.rows, group=self.group,
                                             out=self.gathered, async_op=True)
            if res is not None:
                self.gathered = res
            self.pending[i] = handle
        self.k += 1
        return self.bufs[i]

    def finish(self):
        for i in (0, 1):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None
        return self.gathered
