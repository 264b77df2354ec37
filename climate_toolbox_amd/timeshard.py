"""Time-axis sharding across the GPUs of one node.

Output row t depends only on input row t, so the leading (time / ensemble x time) axis is cut
into `world` contiguous blocks, one process per GPU, no exchange during compute; ONE collective
at the end reassembles the (T, R) region time series on a root rank (RCCL gather over xGMI when
the process group is "nccl"; the same code runs on "gloo" for the CPU tests).
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


def gather_time_shards(out_local, dst=0, rows=None, group=None, out=None):
    """Gather per-rank (T_rank, R) blocks on `dst`; returns the (sum T_rank, R) tensor there and
    None elsewhere.  `rows` = per-rank row counts when they differ (ragged shards are padded to
    the largest block for the collective and trimmed on arrival).  With equal blocks, `out` (a
    contiguous (world * T_rank, R) tensor on `dst`) receives the blocks in place: no temporaries,
    no concatenation."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rows is None:
        rows = [out_local.shape[0]] * world
    if out_local.shape[0] != rows[rank]:
        raise ValueError("rank %d holds %d rows, rows[%d] = %d" % (rank, out_local.shape[0], rank, rows[rank]))
    mx = max(rows)
    block = out_local
    if block.shape[0] != mx:
        block = torch.zeros((mx,) + tuple(out_local.shape[1:]), dtype=out_local.dtype, device=out_local.device)
        block[: out_local.shape[0]] = out_local
    block = block.contiguous()
    if rank == dst and out is not None and all(n == mx for n in rows):
        if tuple(out.shape) != (world * mx,) + tuple(block.shape[1:]) or not out.is_contiguous() or out.dtype != block.dtype:
            raise ValueError("out must be a contiguous %r tensor" % (((world * mx,) + tuple(block.shape[1:])),))
        dist.gather(block, gather_list=[out[i * mx:(i + 1) * mx] for i in range(world)], dst=dst, group=group)
        return out
    if rank == dst:
        parts = [torch.empty_like(block) for _ in range(world)]
        dist.gather(block, gather_list=parts, dst=dst, group=group)
        return torch.cat([p[:n] for p, n in zip(parts, rows)], dim=0)
    dist.gather(block, gather_list=None, dst=dst, group=group)
    return None


def aggregate_time_sharded(apply_fn, X_local, rows=None, dst=0, group=None):
    """apply_fn(X_local) -> (T_rank, R) on this rank's device, then the gather above."""
    return gather_time_shards(apply_fn(X_local), dst=dst, rows=rows, group=group)
