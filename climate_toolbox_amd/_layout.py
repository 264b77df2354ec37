"""How a variable's array flattens to the (time x gridcell) / (gridcell x time) matrix the kernels take, and back
(S10: the group dim takes the slot of the first of lat / lon) -- bookkeeping of shapes; host arrays and torch CUDA tensors.

Split out of aggregations.py in round 6; aggregations.py re-exports every name."""
from __future__ import annotations

import numpy as np

from . import engine as _engine


def _spatial_layout(dims):
    """Where lat/lon sit among the dims -> how the array flattens to (T x G) or (G x T)."""
    dims = tuple(dims)
    if "lat" not in dims or "lon" not in dims:
        raise KeyError("dataset must have dimensions named 'lat' and 'lon' (aggregations.py:27), "
                       "got %r" % (dims,))
    ia, io = dims.index("lat"), dims.index("lon")
    first, second = (ia, io) if ia < io else (io, ia)
    others = [i for i in range(len(dims)) if i not in (ia, io)]
    return ia, io, first, second, others


def _result_dims(dims, group_dim):
    """S10: the group dim takes the slot of the first indexed dim; the second one disappears."""
    _, _, first, second, _ = _spatial_layout(dims)
    return tuple(group_dim if i == first else d for i, d in enumerate(dims) if i != second)


def _is_device_tensor(values):
    """A torch CUDA tensor handed in as a variable's buffer: the field is already in HBM."""
    return type(values).__module__.startswith("torch") and getattr(values, "is_cuda", False)


def _to_device(X2):
    """2-D host array -> device tensor (``wagg_upload``: one DMA from the array page-locked in place for the call -- no
    pageable pointer goes to a runtime copy, no pin is left behind); device tensors pass through."""
    if _is_device_tensor(X2):
        return X2
    return _engine.upload(X2)


def _flatten_for_device(values, dims):
    """(values, dims) -> (2-D C-contiguous array, layout, others_shape, unflatten(result2d, R)).
    ``values`` is a NumPy array or a torch CUDA tensor (then the re-layout, if any, runs on the
    device and nothing crosses PCIe)."""
    on_dev = _is_device_tensor(values)
    if on_dev:
        import torch
        if values.dtype not in (torch.float32, torch.float64):
            values = _engine.to_float64(values)      # the reference's promotion (S8), in the library's own kernel
        # the re-layout of a device field runs in the library's own kernel (wagg_relayout_*), not in a torch one
        contig = lambda a: a if a.is_contiguous() else _engine.relayout(a)
        transpose = lambda a, order: _engine.relayout(a, order)
    else:
        values = np.asarray(values)
        if values.dtype not in (np.float32, np.float64):
            values = values.astype(np.float64)
        contig, transpose = np.ascontiguousarray, np.transpose
    ia, io, first, second, others = _spatial_layout(dims)
    shape = tuple(values.shape)
    G = shape[ia] * shape[io]
    adjacent = second == first + 1
    if adjacent and all(i < first for i in others):          # (..., lat, lon): gridcell axis contiguous
        T = int(np.prod([shape[i] for i in others])) if others else 1
        X2 = contig(values).reshape(T, G)
        layout = "TG"
    elif adjacent and all(i > second for i in others):       # (lat, lon, ...): the test fixture
        T = int(np.prod([shape[i] for i in others])) if others else 1
        X2 = contig(values).reshape(G, T)
        layout = "GT"
    else:                                                    # anything else: one transpose
        order = others + [first, second]
        X2 = contig(transpose(values, order)).reshape(-1, G)
        layout = "TG"
    others_shape = tuple(shape[i] for i in others)
    n_before = sum(1 for i in others if i < first)

    def unflatten(res2d, R):
        # (NumPy arrays, or -- under results_on_device() -- torch CUDA tensors: views only, nothing moves)
        moveaxis = np.moveaxis if isinstance(res2d, np.ndarray) else (lambda t, src, dst: t.movedim(src, dst))
        if layout == "TG":
            arr = res2d.reshape(others_shape + (R,))
        else:
            arr = moveaxis(res2d.reshape((R,) + others_shape), 0, -1)
        return moveaxis(arr, -1, n_before)

    return X2, layout, others_shape, unflatten
