"""The page-locked result pool of the drop-in: device results return to the host through blocks of this pool (one DMA at
PCIe speed), and host-resident applies write their results straight into them.

Split out of aggregations.py in round 6; aggregations.py re-exports the names (the cap is set HERE: ``_pinned._PINNED_OUT_CAP``)."""
from __future__ import annotations

import numpy as np

from ._memo import _CACHE_LOCK

# Results go back to the host through page-locked memory (one DMA at PCIe speed instead of the runtime's staged
# pageable copy: 36 MB in 0.8 instead of 3.6 ms).  The blocks come from a small pool of the module's own (round 4): a block
# returns to the pool when the caller drops the result and is handed out again as it is -- page-locking 36 MB anew costs
# 3-4 ms, and whether torch's caching host allocator had a recycled block ready decided between a 3 ms and an 8 ms call.
# The amount of pooled memory is capped; beyond it (and for small results) the pageable copy is used.
_PINNED_POOL = {"bytes": 0, "free": {}, "lru": []}  # free: rounded size -> [uint8 pinned tensors]; lru: sizes, least recently used first
_PINNED_OUT_CAP = 512 << 20


def _pinned_return(block):
    with _CACHE_LOCK:
        _PINNED_POOL["free"].setdefault(block.numel(), []).append(block)


def _pinned_touch(size):
    lru = _PINNED_POOL["lru"]
    if size in lru:
        lru.remove(size)
    lru.append(size)


def _pinned_make_room(size):
    """Pool at its cap: give back FREE blocks of other sizes, least recently used size first, until ``size`` more bytes fit
    (a workload whose result shapes change would otherwise fill the pool with blocks nobody asks for again and send every
    new shape down the pageable copy for good).  Caller holds _CACHE_LOCK; returns True when the bytes fit now."""
    free = _PINNED_POOL["free"]
    dropped = False
    for other in [z for z in _PINNED_POOL["lru"] if z != size] + [z for z in list(free) if z != size and z not in _PINNED_POOL["lru"]]:
        blocks = free.get(other)
        while blocks and _PINNED_POOL["bytes"] + size > _PINNED_OUT_CAP:
            blocks.pop()                                             # (the storage goes back to torch's host allocator ...)
            _PINNED_POOL["bytes"] -= other
            dropped = True
        if _PINNED_POOL["bytes"] + size <= _PINNED_OUT_CAP:
            break
    if dropped:
        _release_host_cache()                                        # (... which only now unpins and frees it)
    return _PINNED_POOL["bytes"] + size <= _PINNED_OUT_CAP


def _release_host_cache():
    """torch's caching HOST allocator keeps a dropped pinned block page-locked in its own cache: without this the
    pool's cap would bound the pool's accounting, not the process's page-locked memory.  Its empty-cache hook is private
    (``torch._C._host_emptyCache``); where a torch build lacks it the cap is accounting only."""
    import torch
    hook = getattr(torch._C, "_host_emptyCache", None)
    if hook is not None:
        try:
            hook()
        except RuntimeError:
            pass


def _pinned_result(shape, dtype):
    """An uninitialised host array of ``shape`` / ``dtype`` in a page-locked block of the result pool (returned to the pool
    when the array and every view of it are gone), or None: the array is small, the pool is at its cap with every block in
    use, or no page-locked memory can be had."""
    import weakref
    import torch
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if n < (1 << 20):
        return None
    size = (n + (1 << 20) - 1) >> 20 << 20                          # blocks of whole MiB: results of one shape share them
    block = None
    with _CACHE_LOCK:
        free = _PINNED_POOL["free"].get(size)
        if free:
            block = free.pop()
        elif _PINNED_POOL["bytes"] + size <= _PINNED_OUT_CAP or (size <= _PINNED_OUT_CAP and _pinned_make_room(size)):
            _PINNED_POOL["bytes"] += size
            block = False                                            # allocate below, outside the lock
        if block is not None:
            _pinned_touch(size)
    if block is None:
        return None                                                  # the pool is at its cap and every block is in use
    if block is False:
        try:
            block = torch.empty(size, dtype=torch.uint8, pin_memory=True)
        except RuntimeError:                                         # no page-locked memory to be had
            with _CACHE_LOCK:
                _PINNED_POOL["bytes"] -= size
            return None
    arr = block.numpy()[:n].view(dtype).reshape(shape)              # shares the block
    root = arr
    while isinstance(root.base, np.ndarray):      # the ndarray every view of this memory keeps alive (numpy collapses chains
        root = root.base                          # of views onto it): the finalizer goes there, not onto an intermediate view
    weakref.finalize(root, _pinned_return, block)  # the finalizer holds the block: it outlives every view of the result
    return arr


def _to_host(o):
    """Device tensor -> host array, through a page-locked block of the result pool when one can be had."""
    import torch
    arr = _pinned_result(tuple(o.shape), str(o.dtype).replace("torch.", "")) if o.numel() else None
    if arr is None:
        return o.cpu().numpy()
    torch.from_numpy(arr).copy_(o)
    return arr


def _clear_free_blocks():
    """Give the FREE blocks of the pool back (clear_caches); blocks behind results the caller still holds stay."""
    with _CACHE_LOCK:
        # (a snapshot: a result dropped while we are here returns its block through _pinned_return -- same thread, the lock is
        #  re-entrant -- and may add a size to the dict)
        for size, blocks in list(_PINNED_POOL["free"].items()):
            _PINNED_POOL["bytes"] -= size * len(blocks)
            blocks.clear()
        _PINNED_POOL["lru"].clear()
    _release_host_cache()                       # (torch's host allocator keeps dropped pinned blocks: really free them)
