"""Device-side objects over the C-ABI (include/wagg.h).  torch supplies device memory and streams
only; every number is produced by the HIP kernels in climate_toolbox_amd/csrc/."""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import LAYOUT_GT, LAYOUT_TG, OUT_RT, OUT_TR, WaggError

_LAYOUTS = {"TG": LAYOUT_TG, "GT": LAYOUT_GT}
_OUTS = {"TR": OUT_TR, "RT": OUT_RT}


def require_gpu():
    L = _lib.load()
    if L.wagg_device_count() < 1:
        raise WaggError("no HIP device visible: the aggregation engine has no CPU fallback")
    import torch
    if not torch.cuda.is_available():
        raise WaggError("torch sees no GPU (needed for device buffers and streams)")
    return torch


class _on_device:
    """``with _on_device(d):`` -- device ``d`` is current inside (None: leave it as it is)."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        if self.device is not None:
            import torch
            self._ctx = torch.cuda.device(int(self.device))
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.device is not None:
            self._ctx.__exit__(*exc)
        return False


def _current_device():
    import torch
    return int(torch.cuda.current_device())


def _stream_handle(stream):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _elem(dtype):
    """WAGG_T_F32 / WAGG_T_F64 of a torch or numpy dtype"""
    return _lib.T_F64 if str(dtype).endswith("float64") else _lib.T_F32


def _ld(t):
    """Leading dimension of a 2-D row-contiguous tensor: the row pitch; a single row carries an
    arbitrary stride(0), so its own length stands in."""
    return max(int(t.stride(0)), int(t.shape[1]), 1) if t.shape[0] > 1 else max(1, int(t.shape[1]))


def _np_ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _host_out(out, shape, dtype):
    """The result array of a host form: the caller's (C-contiguous, right shape and dtype -- e.g. a page-locked block,
    which the library then uses as it is) or a fresh one."""
    if out is None:
        return np.empty(shape, dtype=dtype)
    if not (isinstance(out, np.ndarray) and tuple(out.shape) == tuple(shape) and out.dtype == np.dtype(dtype) and out.flags.c_contiguous
            and out.flags.writeable):
        raise ValueError("out must be a writeable C-contiguous %s array of shape %s" % (np.dtype(dtype), tuple(shape)))
    return out


def _check_X(X, layout):
    import torch
    if not (isinstance(X, torch.Tensor) and X.is_cuda and X.dim() == 2):
        raise TypeError("X must be a 2-D CUDA torch tensor")
    if X.dtype not in (torch.float32, torch.float64):
        raise TypeError("X must be float32 or float64, got %s" % X.dtype)
    if X.shape[1] > 1 and X.stride(1) != 1:
        raise ValueError("X rows must be contiguous (stride(1) == 1)")
    if layout not in _LAYOUTS:
        raise ValueError("layout must be 'TG' or 'GT'")
    return X


class SparsePlan:
    """Coded segment table -> device plan (wagg_plan_create).  Immutable after construction."""

    def __init__(self, cell_idx, region_code, w_eff, G, R, row_len=0, flags=0, device=None):
        require_gpu()
        L = _lib.load()
        self._args = (cell_idx, region_code, w_eff, G, R, row_len, flags)      # what a replica on another device is built from
        ci = np.ascontiguousarray(cell_idx, dtype=np.int32)
        rc = np.ascontiguousarray(region_code, dtype=np.int32)
        we = np.ascontiguousarray(w_eff, dtype=np.float64)
        if not (ci.shape == rc.shape == we.shape and ci.ndim == 1):
            raise ValueError("cell_idx, region_code, w_eff must be 1-D and of equal length")
        self._h = C.c_void_p()
        self._lease = threading.Lock()       # held by whoever is applying a cached plan (aggregations._plan_for)
        self.G, self.R = int(G), int(R)
        with _on_device(device):
            self.device = _current_device()
            _lib.check(L.wagg_plan_create(_np_ptr(ci, C.c_int32), _np_ptr(rc, C.c_int32),
                                          _np_ptr(we, C.c_double), len(ci), self.G, self.R,
                                          int(row_len), int(flags), C.byref(self._h)), "wagg_plan_create")
        info = _lib.PlanInfo()
        _lib.check(L.wagg_plan_get_info_sized(self._h, C.byref(info), C.sizeof(info)), "wagg_plan_get_info_sized")
        self.info = {k: getattr(info, k) for k, _ in _lib.PlanInfo._fields_}
        den = np.empty(self.R, dtype=np.float64)
        _lib.check(L.wagg_plan_get_den(self._h, _np_ptr(den, C.c_double)), "wagg_plan_get_den")
        self.den = den

    def close(self):
        for r in getattr(self, "_replicas", {}).values():
            r.close()
        if getattr(self, "_h", None) is not None and self._h.value and _lib is not None and _lib._lib is not None:
            _lib._lib.wagg_plan_destroy(self._h)     # (module globals may be gone at interpreter exit)
            self._h = C.c_void_p()

    __del__ = close

    def status(self, stream=None):
        """Synchronise the stream and raise if any apply on this plan failed on the device
        (``wagg_plan_status``)."""
        _lib.check(_lib.load().wagg_plan_status(self._h, _stream_handle(stream)), "wagg_plan_status")

    def apply(self, X, layout="TG", out=None, out_layout="TR", stream=None):
        """out[t, r] on the device; asynchronous on torch's current stream (or `stream`)."""
        import torch
        X = _check_X(X, layout)
        T = X.shape[0] if layout == "TG" else X.shape[1]
        n_g = X.shape[1] if layout == "TG" else X.shape[0]
        if n_g != self.G:
            raise ValueError("X has %d grid cells, plan expects %d" % (n_g, self.G))
        shape = (T, self.R) if out_layout == "TR" else (self.R, T)
        if out is None:
            out = torch.empty(shape, dtype=X.dtype, device=X.device)
        elif tuple(out.shape) != shape or out.dtype != X.dtype or (shape[1] > 1 and out.stride(1) != 1):
            raise ValueError("out must be a %s %s tensor with contiguous rows" % (shape, X.dtype))
        _lib.run("wagg_apply", plan_kind=_lib.PLAN_SEGMENT, plan=self._h, elem=_elem(X.dtype), source=_lib.SRC_DEVICE,
                 x=X.data_ptr(), T=T, ldx=_ld(X), layout=_LAYOUTS[layout], out=out.data_ptr(), ldo=_ld(out),
                 out_layout=_OUTS[out_layout], stream=_stream_handle(stream))
        return out

    def apply_poly(self, X, offset, n_pow, layout="TG", out=None, out_layout="TR", stream=None, pow_first=1):
        """out[i, t, r] = aggregate of (X + offset)**(pow_first + i), i < n_pow (``wagg_apply_poly_*``:
        the arithmetic of tas_poly, transformations.py:188, fused into the loads of the aggregation)."""
        import torch
        X = _check_X(X, layout)
        T = X.shape[0] if layout == "TG" else X.shape[1]
        n_g = X.shape[1] if layout == "TG" else X.shape[0]
        if n_g != self.G:
            raise ValueError("X has %d grid cells, plan expects %d" % (n_g, self.G))
        shape = (int(n_pow),) + ((T, self.R) if out_layout == "TR" else (self.R, T))
        if out is None:
            out = torch.empty(shape, dtype=X.dtype, device=X.device)
        elif tuple(out.shape) != shape or out.dtype != X.dtype or not out.is_contiguous():
            raise ValueError("out must be a contiguous %s %s tensor" % (shape, X.dtype))
        _lib.run("wagg_apply (poly)", plan_kind=_lib.PLAN_SEGMENT, plan=self._h, elem=_elem(X.dtype), source=_lib.SRC_DEVICE,
                 transform=_lib.XF_POLY, offset=float(offset), pow_first=int(pow_first), n_pow=int(n_pow),
                 x=X.data_ptr(), T=T, ldx=_ld(X), layout=_LAYOUTS[layout], out=out.data_ptr(), ldo=max(1, shape[2]),
                 out_pstride=shape[1] * shape[2], out_layout=_OUTS[out_layout], stream=_stream_handle(stream))
        return out

    def apply_edd(self, tasmin, tasmax, thresholds, offset=0.0, layout="TG", out=None, out_layout="TR", stream=None):
        """out[i, t, r] = aggregate of snyder_edd(tasmin + offset, tasmax + offset, thresholds[i])
        (``wagg_apply_edd_*``; transformations.py:7-93 evaluated while the two fields are loaded)."""
        import torch
        tasmin, tasmax = _check_X(tasmin, layout), _check_X(tasmax, layout)
        if tasmin.shape != tasmax.shape or tasmin.dtype != tasmax.dtype or _ld(tasmin) != _ld(tasmax):
            raise ValueError("tasmin and tasmax must have the same shape, dtype and row stride")
        T = tasmin.shape[0] if layout == "TG" else tasmin.shape[1]
        n_g = tasmin.shape[1] if layout == "TG" else tasmin.shape[0]
        if n_g != self.G:
            raise ValueError("fields have %d grid cells, plan expects %d" % (n_g, self.G))
        thr = np.ascontiguousarray(np.atleast_1d(thresholds), dtype=np.float64)
        shape = (len(thr),) + ((T, self.R) if out_layout == "TR" else (self.R, T))
        if out is None:
            out = torch.empty(shape, dtype=tasmin.dtype, device=tasmin.device)
        elif tuple(out.shape) != shape or out.dtype != tasmin.dtype or not out.is_contiguous():
            raise ValueError("out must be a contiguous %s %s tensor" % (shape, tasmin.dtype))
        _lib.run("wagg_apply (edd)", plan_kind=_lib.PLAN_SEGMENT, plan=self._h, elem=_elem(tasmin.dtype), source=_lib.SRC_DEVICE,
                 transform=_lib.XF_EDD, offset=float(offset), thresholds=thr.ctypes.data, n_thr=len(thr),
                 x=tasmin.data_ptr(), x2=tasmax.data_ptr(), T=T, ldx=_ld(tasmin), layout=_LAYOUTS[layout], out=out.data_ptr(),
                 ldo=max(1, shape[2]), out_pstride=shape[1] * shape[2], out_layout=_OUTS[out_layout], stream=_stream_handle(stream))
        return out

    def replica(self, device):
        """The same table as a plan on another device (multi-device host streaming)."""
        return SparsePlan(*self._args, device=device)

    def apply_host(self, X, layout="TG", out_layout="TR", flags=0, replicas=(), out=None):
        """Blocking host-buffer form (wagg_apply_host_ex_*): numpy in, numpy out.  (time, gridcell) data
        is streamed through the device in row blocks (H2D of block i+1, the kernels of block i and the
        return of block i-1 at once); ``flags``: ``_lib.HOST_PIN`` page-locks the arrays for the call,
        ``_lib.HOST_WHOLE`` copies the whole field at once, ``_lib.HOST_LINES`` lets host threads pack the 128-byte
        lines the table references so that only those cross PCIe (one device; ignored with ``replicas``).  ``replicas``: further plans of the same table
        on other devices (``replica(d)``): the row blocks are then dealt over all of them
        (``wagg_apply_host_multi_*``), each device on its own PCIe link."""
        X = np.ascontiguousarray(X)
        if X.dtype not in (np.float32, np.float64) or X.ndim != 2:
            raise TypeError("X must be a 2-D float32/float64 array")
        T = X.shape[0] if layout == "TG" else X.shape[1]
        shape = (T, self.R) if out_layout == "TR" else (self.R, T)
        out = _host_out(out, shape, X.dtype)
        if replicas:
            if layout != "TG" or out_layout != "TR":
                raise ValueError("the multi-device form takes (time, gridcell) data only")
            plans = (self,) + tuple(replicas)
            hs = (C.c_void_p * len(plans))(*[p._h for p in plans])
            devs = (C.c_int32 * len(plans))(*[p.device for p in plans])
            _lib.run("wagg_apply (host, multi-device)", plan_kind=_lib.PLAN_SEGMENT, plan=hs, n_plans=len(plans), devices=devs,
                     elem=_elem(X.dtype), source=_lib.SRC_HOST_MULTI, x=X.ctypes.data, T=T, ldx=X.shape[1], out=out.ctypes.data,
                     ldo=max(1, self.R), flags=int(flags) & ~(_lib.HOST_LINES | _lib.HOST_LINES_WHOLE))
            return out
        with _on_device(self.device):
            _lib.run("wagg_apply (host)", plan_kind=_lib.PLAN_SEGMENT, plan=self._h, elem=_elem(X.dtype), source=_lib.SRC_HOST,
                     x=X.ctypes.data, T=T, ldx=X.shape[1], layout=_LAYOUTS[layout], out=out.ctypes.data, ldo=max(1, shape[1]),
                     out_layout=_OUTS[out_layout], flags=int(flags))
        return out


    def apply_poly_host(self, X, offset, n_pow, pow_first=1, flags=0, out=None):
        """The fused powers of a host-resident (time, gridcell) field (``wagg_apply_poly_host_*``): numpy in, a
        (n_pow, T, R) numpy array out; the field crosses PCIe once (``_lib.HOST_LINES``: only the lines the table
        references), every row block is raised to its powers on the device."""
        X = np.ascontiguousarray(X)
        if X.dtype not in (np.float32, np.float64) or X.ndim != 2:
            raise TypeError("X must be a 2-D float32/float64 array")
        if X.shape[1] != self.G:
            raise ValueError("X has %d grid cells, plan expects %d" % (X.shape[1], self.G))
        T = X.shape[0]
        out = _host_out(out, (int(n_pow), T, self.R), X.dtype)
        with _on_device(self.device):
            _lib.run("wagg_apply (host, poly)", plan_kind=_lib.PLAN_SEGMENT, plan=self._h, elem=_elem(X.dtype), source=_lib.SRC_HOST,
                     transform=_lib.XF_POLY, offset=float(offset), pow_first=int(pow_first), n_pow=int(n_pow), x=X.ctypes.data, T=T,
                     ldx=X.shape[1], out=out.ctypes.data, ldo=max(1, self.R), out_pstride=max(1, T * self.R), flags=int(flags))
        return out


    def apply_edd_host(self, tasmin, tasmax, thresholds, offset=0.0, flags=0, out=None):
        """Snyder degree days of two host-resident (time, gridcell) fields at each threshold, aggregated
        (``wagg_apply_edd_host_*``): numpy in, a (n_thr, T, R) numpy array out; both fields cross PCIe once
        (``_lib.HOST_LINES``: only the lines the table references), the formula runs on the device."""
        tasmin, tasmax = np.ascontiguousarray(tasmin), np.ascontiguousarray(tasmax)
        if tasmin.dtype not in (np.float32, np.float64) or tasmin.ndim != 2:
            raise TypeError("fields must be 2-D float32/float64 arrays")
        if tasmin.shape != tasmax.shape or tasmin.dtype != tasmax.dtype:
            raise ValueError("tasmin and tasmax must have the same shape and dtype")
        if tasmin.shape[1] != self.G:
            raise ValueError("fields have %d grid cells, plan expects %d" % (tasmin.shape[1], self.G))
        thr = np.ascontiguousarray(np.atleast_1d(thresholds), dtype=np.float64)
        T = tasmin.shape[0]
        out = _host_out(out, (len(thr), T, self.R), tasmin.dtype)
        with _on_device(self.device):
            _lib.run("wagg_apply (host, edd)", plan_kind=_lib.PLAN_SEGMENT, plan=self._h, elem=_elem(tasmin.dtype), source=_lib.SRC_HOST,
                     transform=_lib.XF_EDD, offset=float(offset), thresholds=thr.ctypes.data, n_thr=len(thr), x=tasmin.ctypes.data,
                     x2=tasmax.ctypes.data, T=T, ldx=tasmin.shape[1], out=out.ctypes.data, ldo=max(1, self.R),
                     out_pstride=max(1, T * self.R), flags=int(flags))
        return out


class DensePlan:
    """Dense-family plan: W as a (gridcell x region) matrix resident in HBM contracted on the matrix
    cores (full or tile-sparse form; fp32 or fp64 weights), or per-wave entry lists for scattered,
    sparse weights (fp32).  ``info["form"]`` says which (``_lib.FORM_*``), ``dtype`` the element type
    the plan serves ("float32" / "float64")."""

    def __init__(self, handle, G, R, device=None, recipe=None):
        self._h, self.G, self.R = handle, int(G), int(R)
        self.device = _current_device() if device is None else int(device)
        self._recipe = recipe                # synthetic plans: (constructor, args, kwargs) a replica is generated from
        self._lease = threading.Lock()
        den = np.empty(self.R, dtype=np.float64)
        _lib.check(_lib.load().wagg_dense_get_den(self._h, _np_ptr(den, C.c_double)), "wagg_dense_get_den")
        self.den = den
        inf = _lib.DenseInfo()
        _lib.check(_lib.load().wagg_dense_get_info_sized(self._h, C.byref(inf), C.sizeof(inf)), "wagg_dense_get_info_sized")
        self.info = {k: (float if ct is C.c_double else int)(getattr(inf, k)) for k, ct in _lib.DenseInfo._fields_}
        self.dtype = "float64" if self.info["elem_bytes"] == 8 else "float32"

    @staticmethod
    def _is64(dtype):
        return str(dtype).endswith("float64") or dtype is np.float64

    @classmethod
    def synth_blocklocal(cls, G, R, seed, fill=0.952, dtype="float32", device=None):
        """c5's block-local weights, generated on the device in tile-sparse form."""
        require_gpu()
        h = C.c_void_p()
        L = _lib.load()
        fn = L.wagg_dense_create_synth_blocklocal_f64 if cls._is64(dtype) else L.wagg_dense_create_synth_blocklocal
        with _on_device(device):
            dev = _current_device()
            _lib.check(fn(int(G), int(R), int(seed), float(fill), C.byref(h)), "wagg_dense_create_synth_blocklocal")
        return cls(h, G, R, dev, (cls.synth_blocklocal, (G, R, seed), dict(fill=fill, dtype=dtype)))

    @classmethod
    def synth(cls, G, R, seed, fill=1.0, dtype="float32", device=None):
        """W[g, r] = hash_u01(g R + r, seed); with fill < 1 only that fraction of the entries, at
        uniformly random positions (c5's uniform-random structure at fill = 0.01: entry lists)."""
        require_gpu()
        h = C.c_void_p()
        L = _lib.load()
        fn = L.wagg_dense_create_synth_f64 if cls._is64(dtype) else L.wagg_dense_create_synth_sparse
        with _on_device(device):
            dev = _current_device()
            _lib.check(fn(int(G), int(R), int(seed), float(fill), C.byref(h)), "wagg_dense_create_synth")
        return cls(h, G, R, dev, (cls.synth, (G, R, seed), dict(fill=fill, dtype=dtype)))

    @classmethod
    def from_host(cls, W, device=None):
        """From a host (G, R) matrix; a float64 array makes an fp64 plan, anything else fp32."""
        require_gpu()
        W = np.asarray(W)
        h = C.c_void_p()
        with _on_device(device):
            dev = _current_device()
            if W.dtype == np.float64:
                W = np.ascontiguousarray(W)
                _lib.check(_lib.load().wagg_dense_create_host_f64(_np_ptr(W, C.c_double), W.shape[0], W.shape[1], C.byref(h)),
                           "wagg_dense_create_host_f64")
            else:
                W = np.ascontiguousarray(W, dtype=np.float32)
                _lib.check(_lib.load().wagg_dense_create_host(_np_ptr(W, C.c_float), W.shape[0], W.shape[1], C.byref(h)),
                           "wagg_dense_create_host")
        return cls(h, W.shape[0], W.shape[1], dev)

    @classmethod
    def from_segments(cls, cell_idx, region_code, w_eff, G, R, dtype="float32", device=None, form=None):
        """From the coded segment table (COO rows; ``wagg_dense_create_from_segments*``).  ``form``: None / "auto" lets the
        library choose (the caller never needs to know the form, like the reference's single weights type,
        aggregations.py:64-73); "full" / "tiles" / "entries" pin it (measurements, tests).  At most 2**31 - 1 rows."""
        require_gpu()
        ci = np.ascontiguousarray(cell_idx, dtype=np.int32)
        rc = np.ascontiguousarray(region_code, dtype=np.int32)
        we = np.ascontiguousarray(w_eff, dtype=np.float64)
        h = C.c_void_p()
        L = _lib.load()
        fn = L.wagg_dense_create_from_segments_f64 if cls._is64(dtype) else L.wagg_dense_create_from_segments
        with _on_device(device):
            dev = _current_device()
            _lib.check(fn(_np_ptr(ci, C.c_int32), _np_ptr(rc, C.c_int32), _np_ptr(we, C.c_double), len(ci), int(G),
                          int(R), _lib.FORCE_FORM[form], C.byref(h)), "wagg_dense_create_from_segments")
        return cls(h, G, R, dev)

    @classmethod
    def from_csr(cls, rowptr, col, val, G, R, dtype="float32", device=None, form=None, general_sort=False):
        """From a caller's table in CSR form (``wagg_dense_create_from_csr*``; BASELINE configs[4] "sparse CSR weights"):
        ``rowptr`` (G + 1 offsets, rows = grid cells), ``col`` = region codes, ``val`` = fp64 weights -- the coded form of
        the reference's weights table (aggregations.py:64-73).  The arrays are uploaded as they are (no host copy when
        they already have the dtypes int64 / int32 / float64); everything else happens on the device.  ``form`` as for
        :meth:`from_segments`.  A table whose columns ascend inside every row (scipy's ``has_sorted_indices``) is put in
        the plan's order by one stable pass per chunk of 128 cells instead of the general radix sort
        (``info["one_pass_sort"]``); ``general_sort=True`` takes the general sort regardless -- same plan, for tests
        and timings."""
        require_gpu()
        rp = np.ascontiguousarray(rowptr, dtype=np.int64)
        co = np.ascontiguousarray(col, dtype=np.int32)
        va = np.ascontiguousarray(val, dtype=np.float64)
        if rp.shape != (int(G) + 1,) or co.shape != va.shape or co.ndim != 1 or (len(rp) and int(rp[-1]) != len(co)):
            raise ValueError("rowptr must hold G + 1 offsets ending at len(col) == len(val)")
        h = C.c_void_p()
        L = _lib.load()
        fn = L.wagg_dense_create_from_csr_f64 if cls._is64(dtype) else L.wagg_dense_create_from_csr
        with _on_device(device):
            dev = _current_device()
            _lib.check(fn(_np_ptr(rp, C.c_int64), _np_ptr(co, C.c_int32), _np_ptr(va, C.c_double), int(G), int(R),
                          _lib.FORCE_FORM[form] | (_lib.DENSE_GENERAL_SORT if general_sort else 0), C.byref(h)),
                       "wagg_dense_create_from_csr")
        return cls(h, G, R, dev)

    def replica(self, device):
        """The same weights as a plan of its own on ``device`` (multi-device host streaming; a dense-family plan
        owns its workspaces, so every pipeline needs its own replica -- also two on one device).  Plans built from a
        table or a host matrix are cloned device to device (``wagg_dense_clone``: the packed plan travels, over xGMI
        between two GPUs -- nothing is uploaded or sorted again and no copy of the caller's arrays is kept for it);
        the synthetic generators run again on the target (a 101 GB operand is quicker made than moved)."""
        if self._recipe is not None:
            fn, args, kw = self._recipe
            return fn(*args, device=device, **kw)
        require_gpu()
        h = C.c_void_p()
        _lib.check(_lib.load().wagg_dense_clone(self._h, int(device), C.byref(h)), "wagg_dense_clone")
        return type(self)(h, self.G, self.R, int(device))

    def close(self):
        for r in getattr(self, "_replicas", {}).values():
            r.close()
        if getattr(self, "_h", None) is not None and self._h.value and _lib is not None and _lib._lib is not None:
            _lib._lib.wagg_dense_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _prep(self, X, out):
        import torch
        X = _check_X(X, "TG")
        want = torch.float64 if self.dtype == "float64" else torch.float32
        if X.dtype != want:
            raise TypeError("this dense plan serves %s data, got %s" % (self.dtype, X.dtype))
        if X.shape[1] != self.G:
            raise ValueError("X has %d grid cells, plan expects %d" % (X.shape[1], self.G))
        T = X.shape[0]
        if out is None:
            out = torch.empty((T, self.R), dtype=want, device=X.device)
        elif tuple(out.shape) != (T, self.R) or out.dtype != want or (self.R > 1 and out.stride(1) != 1):
            raise ValueError("out must be a (%d, %d) %s tensor with contiguous rows" % (T, self.R, self.dtype))
        return X, T, out

    def _run(self, what, **fields):
        _lib.run(what, plan_kind=_lib.PLAN_DENSE, elem=_elem(self.dtype), **fields)

    def apply(self, X, out=None, ksplit=0, stream=None):
        X, T, out = self._prep(X, out)
        self._run("wagg_apply (dense)", plan=self._h, source=_lib.SRC_DEVICE, x=X.data_ptr(), T=T, ldx=_ld(X), out=out.data_ptr(),
                  ldo=_ld(out), ksplit=int(ksplit), stream=_stream_handle(stream))
        return out

    def apply_poly(self, X, offset, power, out=None, ksplit=0, stream=None):
        """Aggregate of (X + offset) ** power (``wagg_dense_apply_poly_*``): the transform of
        tas_poly (transformations.py:188) is evaluated while X is packed."""
        X, T, out = self._prep(X, out)
        self._run("wagg_apply (dense, poly)", plan=self._h, source=_lib.SRC_DEVICE, transform=_lib.XF_POLY, offset=float(offset),
                  pow_first=int(power), n_pow=1, x=X.data_ptr(), T=T, ldx=_ld(X), out=out.data_ptr(), ldo=_ld(out), ksplit=int(ksplit),
                  stream=_stream_handle(stream))
        return out

    def apply_edd(self, tasmin, tasmax, threshold, offset=0.0, out=None, ksplit=0, stream=None):
        """Aggregate of snyder_edd(tasmin + offset, tasmax + offset, threshold)
        (``wagg_dense_apply_edd_*``; transformations.py:64-87 evaluated while the fields are packed)."""
        tasmin, T, out = self._prep(tasmin, out)
        tasmax = _check_X(tasmax, "TG")
        if tasmax.shape != tasmin.shape or tasmax.dtype != tasmin.dtype or _ld(tasmax) != _ld(tasmin):
            raise ValueError("tasmin and tasmax must have the same shape, dtype and row stride")
        thr = (C.c_double * 1)(float(threshold))
        self._run("wagg_apply (dense, edd)", plan=self._h, source=_lib.SRC_DEVICE, transform=_lib.XF_EDD, offset=float(offset),
                  thresholds=thr, n_thr=1, x=tasmin.data_ptr(), x2=tasmax.data_ptr(), T=T, ldx=_ld(tasmin), out=out.data_ptr(),
                  ldo=_ld(out), ksplit=int(ksplit), stream=_stream_handle(stream))
        return out

    def apply_host(self, X, flags=0, replicas=(), out=None):
        """Host-resident (time, gridcell) array through the plan in row blocks (``wagg_dense_apply_host_*``):
        numpy in, numpy out; flags and ``replicas`` as for :meth:`SparsePlan.apply_host` (``_lib.HOST_LINES`` means nothing
        to a dense-family plan -- every cell of a row is an operand -- and is dropped)."""
        flags = int(flags) & ~(_lib.HOST_LINES | _lib.HOST_LINES_WHOLE)
        want = np.float64 if self.dtype == "float64" else np.float32
        X = np.ascontiguousarray(X)
        if X.dtype != want or X.ndim != 2 or X.shape[1] != self.G:
            raise TypeError("X must be a (T, %d) %s array" % (self.G, self.dtype))
        out = _host_out(out, (X.shape[0], self.R), want)
        if replicas:
            plans = (self,) + tuple(replicas)
            hs = (C.c_void_p * len(plans))(*[p._h for p in plans])
            devs = (C.c_int32 * len(plans))(*[p.device for p in plans])
            self._run("wagg_apply (dense, host, multi-device)", plan=hs, n_plans=len(plans), devices=devs, source=_lib.SRC_HOST_MULTI,
                      x=X.ctypes.data, T=X.shape[0], ldx=X.shape[1], out=out.ctypes.data, ldo=max(1, self.R), flags=int(flags))
            return out
        with _on_device(self.device):
            self._run("wagg_apply (dense, host)", plan=self._h, source=_lib.SRC_HOST, x=X.ctypes.data, T=X.shape[0], ldx=X.shape[1],
                      out=out.ctypes.data, ldo=max(1, self.R), flags=int(flags))
        return out

    def saw_inf(self, stream=None):
        """True when an apply since the last call met +-inf in its (transformed) data in one of the MFMA
        forms (``wagg_dense_saw_inf``; synchronises the stream and clears the note)."""
        saw = C.c_int(0)
        _lib.check(_lib.load().wagg_dense_saw_inf(self._h, _stream_handle(stream), C.byref(saw)), "wagg_dense_saw_inf")
        return bool(saw.value)


class ShardGroup:
    """Time-axis shards of ONE job on several devices driven by this process, device-resident data
    (``wagg_shard_group_*`` / ``wagg_*apply_sharded_*``; SURVEY 8b ``n_devices``, 8e): shard i -- a (rows_i, G) tensor on
    ``devices[i]`` -- goes through ``plans[i]`` (a replica of the table on that device) and its block lands directly in its
    rows of the result on ``devices[root]``; the blocks travel by RCCL (grouped send / receive, every shard to the root on its
    own link) or by peer copies (``transport``: "auto", "rccl", "peer"; a device may be listed twice with "peer").  Blocking.
    The process-per-GPU form of the same job is :mod:`climate_toolbox_amd.timeshard` (``torch.distributed``)."""

    _TRANSPORTS = {"auto": _lib.GATHER_AUTO, "rccl": _lib.GATHER_RCCL, "peer": _lib.GATHER_PEER}

    def __init__(self, devices, transport="auto"):
        require_gpu()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        _lib.check(_lib.load().wagg_shard_group_create(arr, len(self.devices), self._TRANSPORTS[transport], C.byref(self._h)),
                   "wagg_shard_group_create")
        n, tr = C.c_int(0), C.c_int(0)
        _lib.check(_lib.load().wagg_shard_group_info(self._h, C.byref(n), C.byref(tr)), "wagg_shard_group_info")
        self.transport = {_lib.GATHER_RCCL: "rccl", _lib.GATHER_PEER: "peer"}[tr.value]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value and _lib._lib is not None:
            _lib._lib.wagg_shard_group_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def apply(self, plans, shards, root=0, out=None):
        """``plans[i]`` / ``shards[i]`` per device of the group; returns the (sum rows, R) tensor on ``devices[root]``."""
        import torch
        n = len(self.devices)
        if len(plans) != n or len(shards) != n:
            raise ValueError("the group has %d shards: need as many plans and tensors" % n)
        dense = isinstance(plans[0], DensePlan)
        if any(isinstance(p, DensePlan) != dense for p in plans):
            raise TypeError("all plans of a call are of one family")
        dtype = shards[0].dtype
        rows, lds = [], set()
        for i, x in enumerate(shards):
            if not (x.is_cuda and x.dim() == 2 and x.dtype == dtype and x.device.index == self.devices[i] and x.shape[1] == plans[0].G):
                raise ValueError("shard %d must be a (rows, %d) %s tensor on device %d" % (i, plans[0].G, dtype, self.devices[i]))
            if x.shape[0] and x.shape[1] > 1 and x.stride(1) != 1:
                raise ValueError("shard %d: rows must be contiguous" % i)
            if x.shape[0] > 1:
                lds.add(_ld(x))
            rows.append(int(x.shape[0]))
        if len(lds) > 1:
            raise ValueError("the shards must share one row stride, got %r" % sorted(lds))
        ldx = lds.pop() if lds else plans[0].G
        R = plans[0].R
        total = sum(rows)
        if out is None:
            out = torch.empty((total, R), dtype=dtype, device="cuda:%d" % self.devices[root])
        elif tuple(out.shape) != (total, R) or out.dtype != dtype or out.device.index != self.devices[root] or (R > 1 and out.stride(1) != 1):
            raise ValueError("out must be a (%d, %d) %s tensor on device %d" % (total, R, dtype, self.devices[root]))
        hp = (C.c_void_p * n)(*[p._h for p in plans])
        xp = (C.c_void_p * n)(*[C.c_void_p(x.data_ptr()) for x in shards])
        rw = (C.c_int64 * n)(*rows)
        for d in set(self.devices):                      # the shards' producers ran on torch's streams: the group uses its own
            torch.cuda.synchronize(d)
        _lib.run("wagg_apply (sharded)", plan_kind=_lib.PLAN_DENSE if dense else _lib.PLAN_SEGMENT, plan=hp, n_plans=n, elem=_elem(dtype),
                 source=_lib.SRC_SHARDED, group=self._h, x=xp, rows=rw, ldx=ldx, out=out.data_ptr(), ldo=_ld(out), root=int(root))
        return out


def gather(X, cell_idx_dev, layout="TG", out_layout="TR", stream=None):
    """Materialised pointwise gather (aggregations.py:27) on the device."""
    import torch
    X = _check_X(X, layout)
    if not (cell_idx_dev.is_cuda and cell_idx_dev.dtype == torch.int32 and cell_idx_dev.is_contiguous()):
        raise TypeError("cell_idx_dev must be a contiguous int32 CUDA tensor")
    T = X.shape[0] if layout == "TG" else X.shape[1]
    n = cell_idx_dev.numel()
    shape = (T, n) if out_layout == "TR" else (n, T)
    out = torch.empty(shape, dtype=X.dtype, device=X.device)
    L = _lib.load()
    fn = L.wagg_gather_f32 if X.dtype == torch.float32 else L.wagg_gather_f64
    _lib.check(fn(C.c_void_p(X.data_ptr()), T, _ld(X), _LAYOUTS[layout],
                  C.c_void_p(cell_idx_dev.data_ptr()), n, C.c_void_p(out.data_ptr()), max(1, shape[1]),
                  _OUTS[out_layout], _stream_handle(stream)), "wagg_gather")
    return out


def _flat_dev(t):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype in (torch.float32, torch.float64)):
        raise TypeError("expected a float32/float64 CUDA tensor")
    return t if t.is_contiguous() else relayout(t)


def transform_poly(X, offset, power, stream=None):
    """(X + offset) ** power elementwise on the device (``wagg_transform_poly_*``; the arithmetic of
    tas_poly, transformations.py:188, as the fused kernels evaluate it)."""
    import torch
    X = _flat_dev(X)
    out = torch.empty_like(X)
    L = _lib.load()
    fn = L.wagg_transform_poly_f32 if X.dtype == torch.float32 else L.wagg_transform_poly_f64
    _lib.check(fn(C.c_void_p(X.data_ptr()), X.numel(), float(offset), int(power), C.c_void_p(out.data_ptr()),
                  _stream_handle(stream)), "wagg_transform_poly")
    return out


def transform_edd(tasmin, tasmax, offset, terms, stream=None):
    """sum_k coef_k * snyder_edd(tasmin + offset, tasmax + offset, threshold_k) elementwise on the
    device (``wagg_transform_edd_*``; transformations.py:64-87, :138-140)."""
    import torch
    lo, hi = _flat_dev(tasmin), _flat_dev(tasmax)
    if lo.shape != hi.shape or lo.dtype != hi.dtype:
        raise ValueError("tasmin and tasmax must have the same shape and dtype")
    coefs = np.ascontiguousarray([c for c, _ in terms], dtype=np.float64)
    thr = np.ascontiguousarray([e for _, e in terms], dtype=np.float64)
    out = torch.empty_like(lo)
    L = _lib.load()
    fn = L.wagg_transform_edd_f32 if lo.dtype == torch.float32 else L.wagg_transform_edd_f64
    _lib.check(fn(C.c_void_p(lo.data_ptr()), C.c_void_p(hi.data_ptr()), lo.numel(), float(offset),
                  _np_ptr(coefs, C.c_double), _np_ptr(thr, C.c_double), len(coefs), C.c_void_p(out.data_ptr()),
                  _stream_handle(stream)), "wagg_transform_edd")
    return out


def combine_planes(stack, coefs, stream=None):
    """sum_k coefs[k] * stack[k] on the device (``wagg_combine_planes_*``): how the aggregated degree days of several
    thresholds become snyder_gdd (transformations.py:138-140).  ``stack``: contiguous (K, ...) CUDA tensor, K <= 8."""
    import torch
    if not (isinstance(stack, torch.Tensor) and stack.is_cuda and stack.is_contiguous() and stack.dtype in (torch.float32, torch.float64)):
        raise TypeError("stack must be a contiguous float32/float64 CUDA tensor")
    cf = np.ascontiguousarray(coefs, dtype=np.float64)
    if len(cf) != stack.shape[0]:
        raise ValueError("one coefficient per plane")
    out = torch.empty(stack.shape[1:], dtype=stack.dtype, device=stack.device)
    L = _lib.load()
    fn = L.wagg_combine_planes_f32 if stack.dtype == torch.float32 else L.wagg_combine_planes_f64
    n = out.numel()
    _lib.check(fn(C.c_void_p(stack.data_ptr()), len(cf), n, _np_ptr(cf, C.c_double), n, C.c_void_p(out.data_ptr()),
                  _stream_handle(stream)), "wagg_combine_planes")
    return out


def take_axis(t, axis, index, stream=None):
    """``t`` with only the positions ``index`` (host integers) kept / re-ordered along ``axis`` (``wagg_take_axis``):
    the leap-day drop and the lon re-ordering of a device-resident field, without a torch kernel."""
    import torch
    if not isinstance(t, torch.Tensor):
        raise TypeError("expected a torch tensor")
    axis = int(axis) % t.dim()
    idx = np.ascontiguousarray(index, dtype=np.int64)
    if len(idx) and (idx.min() < 0 or idx.max() >= t.shape[axis]):
        raise IndexError("index out of range")
    if not t.is_cuda:                                # host tensor: a host copy, nothing for the device to do
        return torch.from_numpy(np.take(t.numpy(), idx, axis=axis))
    if t.dtype not in (torch.float32, torch.float64):
        t = to_float64(t)                            # (rows of 1- and 2-byte elements are not whole words: promote first, S8)
    t = relayout(t) if not t.is_contiguous() else t
    outer = int(np.prod(t.shape[:axis], dtype=np.int64))
    inner_bytes = int(np.prod(t.shape[axis + 1:], dtype=np.int64)) * t.element_size()
    out = torch.empty(tuple(t.shape[:axis]) + (len(idx),) + tuple(t.shape[axis + 1:]), dtype=t.dtype, device=t.device)
    idx_dev = torch.from_numpy(idx).to(t.device)
    _lib.check(_lib.load().wagg_take_axis(C.c_void_p(t.data_ptr()), outer, int(t.shape[axis]), inner_bytes,
                                          C.c_void_p(idx_dev.data_ptr()), len(idx), C.c_void_p(out.data_ptr()),
                                          _stream_handle(stream)), "wagg_take_axis")
    if stream is not None:                           # idx_dev goes back to torch's allocator on return: keep it alive until
        idx_dev.record_stream(stream)                # the kernel on the caller's stream has read it
    return out


def upload(a):
    """A C-contiguous float32 / float64 NumPy array as a CUDA tensor of the current device (``wagg_upload``: page-locked in
    place for one DMA when it is large, through the library's staging pieces otherwise; blocking)."""
    torch = require_gpu()
    a = np.ascontiguousarray(a)
    dt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}.get(a.dtype)
    if dt is None:
        raise TypeError("float32 or float64 arrays only, got %s" % a.dtype)
    out = torch.empty(a.shape, dtype=dt, device="cuda")
    torch.cuda.current_stream().synchronize()          # (the caching allocator may hand out memory the stream still writes)
    _lib.check(_lib.load().wagg_upload(C.c_void_p(out.data_ptr()), C.c_void_p(a.ctypes.data), a.nbytes), "wagg_upload")
    return out


def relayout(t, order=None, stream=None):
    """A contiguous copy of ``t`` (float32/float64 CUDA tensor, any strides), its dims in ``order`` if given
    (``wagg_relayout_*``): what ``permute(order).contiguous()`` would do, in the library's own kernel."""
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype in (torch.float32, torch.float64)):
        raise TypeError("device-resident fields must be float32 or float64 CUDA tensors, got %s" % getattr(t, "dtype", type(t)))
    order = list(range(t.dim())) if order is None else [int(i) for i in order]
    if t.dim() > 6 or t.dim() < 1:
        raise ValueError("1..6 dimensions")
    shape = [int(t.shape[i]) for i in order]
    strides = [int(t.stride(i)) for i in order]
    out = torch.empty(shape, dtype=t.dtype, device=t.device)
    sh = (C.c_int64 * len(shape))(*shape)
    st = (C.c_int64 * len(shape))(*strides)
    L = _lib.load()
    fn = L.wagg_relayout_f32 if t.dtype == torch.float32 else L.wagg_relayout_f64
    _lib.check(fn(C.c_void_p(t.data_ptr()), len(shape), sh, st, C.c_void_p(out.data_ptr()), _stream_handle(stream)), "wagg_relayout")
    return out


_TO_F64_TYPES = {"torch.float16": 0, "torch.bfloat16": 1, "torch.int8": 2, "torch.uint8": 3, "torch.int16": 4, "torch.int32": 5,
                 "torch.int64": 6, "torch.float32": 7, "torch.float64": 8}


def to_float64(t, order=None, stream=None):
    """A contiguous float64 copy of a CUDA tensor of another dtype (any strides; dims in ``order`` if given), converted in
    the library's own kernel (``wagg_relayout_to_f64``) -- the reference's promotion: data times float64 weights is float64."""
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError("expected a CUDA tensor")
    code = _TO_F64_TYPES.get(str(t.dtype))
    if code is None:
        raise TypeError("device-resident fields of dtype %s are not supported" % (t.dtype,))
    order = list(range(t.dim())) if order is None else [int(i) for i in order]
    if t.dim() > 6 or t.dim() < 1:
        raise ValueError("1..6 dimensions")
    shape = [int(t.shape[i]) for i in order]
    strides = [int(t.stride(i)) for i in order]
    out = torch.empty(shape, dtype=torch.float64, device=t.device)
    sh = (C.c_int64 * len(shape))(*shape)
    st = (C.c_int64 * len(shape))(*strides)
    _lib.check(_lib.load().wagg_relayout_to_f64(C.c_void_p(t.data_ptr()), code, len(shape), sh, st, C.c_void_p(out.data_ptr()),
                                                _stream_handle(stream)), "wagg_relayout_to_f64")
    return out


def any_less(a, b, stream=None):
    """True iff some a[i] < b[i] (``wagg_any_less_*``: the tasmax < tasmin check of
    transformations.py:62); blocks."""
    import torch
    a, b = _flat_dev(a), _flat_dev(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError("operands must have the same shape and dtype")
    res = C.c_int(0)
    L = _lib.load()
    fn = L.wagg_any_less_f32 if a.dtype == torch.float32 else L.wagg_any_less_f64
    _lib.check(fn(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), a.numel(), C.byref(res), _stream_handle(stream)),
               "wagg_any_less")
    return bool(res.value)


def to_device(values):
    """NumPy array (one pageable H2D copy) or CUDA tensor (as it is) -> CUDA tensor."""
    import torch
    if isinstance(values, torch.Tensor):
        return values if values.is_cuda else values.cuda()
    return torch.from_numpy(np.ascontiguousarray(values)).cuda()


def synth_field(T, G, seed, base, amp, dtype="float32", device="cuda"):
    """X[t, g] = base + amp * (hash_u01(t*G + g, seed) - 0.5), generated on the device."""
    torch = require_gpu()
    dt = torch.float32 if dtype in ("float32", torch.float32) else torch.float64
    X = torch.empty((T, G), dtype=dt, device=device)
    L = _lib.load()
    fn = L.wagg_synth_field_f32 if dt == torch.float32 else L.wagg_synth_field_f64
    _lib.check(fn(C.c_void_p(X.data_ptr()), T, G, G, int(seed), base, amp, _stream_handle(None)),
               "wagg_synth_field")
    return X


def synth_table_csr(G, R, seed, fill, blocklocal=False):
    """The c5 weight tables of :meth:`DensePlan.synth` / :meth:`DensePlan.synth_blocklocal` as host CSR arrays
    ``(rowptr int64, col int32, val float64)`` -- what a caller with a real table would hand to
    :meth:`DensePlan.from_csr` (``wagg_synth_table_csr``: generated on the device, copied out)."""
    require_gpu()
    L = _lib.load()
    rowptr = np.empty(int(G) + 1, dtype=np.int64)
    nnz = C.c_int64(0)
    args = (int(G), int(R), int(seed), float(fill), 1 if blocklocal else 0, _np_ptr(rowptr, C.c_int64))
    _lib.check(L.wagg_synth_table_csr(*args, None, None, 0, C.byref(nnz)), "wagg_synth_table_csr")
    col = np.empty(nnz.value, dtype=np.int32)
    val = np.empty(nnz.value, dtype=np.float64)
    _lib.check(L.wagg_synth_table_csr(*args, _np_ptr(col, C.c_int32), _np_ptr(val, C.c_double), nnz.value, C.byref(nnz)),
               "wagg_synth_table_csr")
    return rowptr, col, val


def profile_enable(on=True):
    """Record HIP event pairs around the dominant kernel of every following apply."""
    _lib.check(_lib.load().wagg_profile_enable(1 if on else 0), "wagg_profile_enable")


PROFILE_SLOTS = 1024      # wagg.h WAGG_PROFILE_SLOTS


def profile_read():
    """Durations (ms) of the dominant kernels recorded since profile_enable(); blocks."""
    buf = (C.c_float * PROFILE_SLOTS)()
    n = C.c_int(0)
    _lib.check(_lib.load().wagg_profile_read(buf, PROFILE_SLOTS, C.byref(n)), "wagg_profile_read")
    return [float(buf[i]) for i in range(n.value)]


def profile_event_overhead(n=64, stream=None):
    """(median, minimum) milliseconds that an event pair handed to the launch of an EMPTY kernel reads
    (``wagg_profile_event_overhead``): the floor under every ``profile_read`` figure."""
    med, mn = C.c_float(0.0), C.c_float(0.0)
    _lib.check(_lib.load().wagg_profile_event_overhead(_stream_handle(stream), int(n), C.byref(med), C.byref(mn)),
               "wagg_profile_event_overhead")
    return float(med.value), float(mn.value)
