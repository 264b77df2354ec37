"""ctypes binding of libwagg.so (include/wagg.h).  There is NO fallback: if the HIP library is
missing or a call fails, this raises -- the product path never computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwagg.so")

EKEY, EINTERNAL = -6, -7
PLAN_NO_LC, PLAN_NO_STREAM, PLAN_NO_LINES, PLAN_LC_MFMA, PLAN_SERIAL_BUILD = 1, 2, 4, 8, 16
FORM_FULL, FORM_TILES, FORM_ENTRIES = 0, 1, 2
FORCE_FORM = {None: 0, "auto": 0, "full": 1, "tiles": 2, "entries": 3}      # WAGG_DENSE_FORCE_*
HOST_PIN, HOST_WHOLE, HOST_LINES, HOST_LINES_WHOLE = 1, 2, 4, 8
DENSE_GENERAL_SORT = 16        # WAGG_DENSE_GENERAL_SORT
GATHER_AUTO, GATHER_RCCL, GATHER_PEER = 0, 1, 2
LAYOUT_TG, LAYOUT_GT = 0, 1
OUT_TR, OUT_RT = 0, 1

# every symbol include/wagg.h declares (tests check that the .so exports all of them)
EXPORTS = (
    "wagg_version", "wagg_device_count", "wagg_shard_rows", "wagg_last_error", "wagg_profile_enable", "wagg_profile_read",
    "wagg_profile_event_overhead",
    "wagg_resolve_cells", "wagg_backup_fill", "wagg_relabel", "wagg_factorize_i64", "wagg_factorize_bytes",
    "wagg_plan_create", "wagg_plan_destroy", "wagg_plan_get_info", "wagg_plan_get_den", "wagg_plan_status",
    "wagg_apply_f32", "wagg_apply_f64", "wagg_apply_host_f32", "wagg_apply_host_f64",
    "wagg_apply_host_ex_f32", "wagg_apply_host_ex_f64", "wagg_dense_apply_host_f32", "wagg_dense_apply_host_f64",
    "wagg_apply_poly_f32", "wagg_apply_poly_f64", "wagg_apply_edd_f32", "wagg_apply_edd_f64",
    "wagg_apply_poly_host_f32", "wagg_apply_poly_host_f64", "wagg_apply_edd_host_f32", "wagg_apply_edd_host_f64",
    "wagg_gather_f32", "wagg_gather_f64",
    "wagg_transform_poly_f32", "wagg_transform_poly_f64", "wagg_transform_edd_f32", "wagg_transform_edd_f64",
    "wagg_any_less_f32", "wagg_any_less_f64",
    "wagg_dense_create_synth", "wagg_dense_create_host", "wagg_dense_create_from_segments", "wagg_dense_create_synth_blocklocal", "wagg_dense_create_synth_sparse", "wagg_dense_get_info",
    "wagg_dense_destroy", "wagg_dense_clone", "wagg_release_scratch", "wagg_scratch_bytes", "wagg_dense_get_den", "wagg_dense_apply_f32", "wagg_dense_apply_poly_f32",
    "wagg_dense_apply_edd_f32", "wagg_dense_saw_inf",
    "wagg_dense_create_synth_f64", "wagg_dense_create_host_f64", "wagg_dense_create_from_segments_f64",
    "wagg_dense_create_synth_blocklocal_f64", "wagg_dense_apply_f64", "wagg_dense_apply_poly_f64", "wagg_dense_apply_edd_f64",
    "wagg_synth_field_f32", "wagg_synth_field_f64",
    "wagg_apply_host_multi_f32", "wagg_apply_host_multi_f64", "wagg_dense_apply_host_multi_f32", "wagg_dense_apply_host_multi_f64",
    "wagg_host_block_plan", "wagg_host_stats_read",
    "wagg_combine_planes_f32", "wagg_combine_planes_f64", "wagg_take_axis", "wagg_relayout_f32", "wagg_relayout_f64",
    "wagg_dense_create_from_csr", "wagg_dense_create_from_csr_f64", "wagg_synth_table_csr", "wagg_relayout_to_f64", "wagg_upload",
    "wagg_shard_group_create", "wagg_shard_group_destroy", "wagg_shard_group_info",
    "wagg_apply_sharded_f32", "wagg_apply_sharded_f64", "wagg_dense_apply_sharded_f32", "wagg_dense_apply_sharded_f64",
    "wagg_apply", "wagg_struct_size", "wagg_struct_ordinals", "wagg_plan_get_info_sized", "wagg_dense_get_info_sized",
    "wagg_host_stats_read_sized",
)
STRUCT_PLAN_INFO, STRUCT_DENSE_INFO, STRUCT_HOST_STATS, STRUCT_APPLY_DESC = 0, 1, 2, 3
PLAN_SEGMENT, PLAN_DENSE = 0, 1
SRC_DEVICE, SRC_HOST, SRC_HOST_MULTI, SRC_SHARDED = 0, 1, 2, 3
XF_NONE, XF_POLY, XF_EDD = 0, 1, 2
T_F32, T_F64 = 7, 8


class HostStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("calls", "blocks", "registered", "register_failed", "unregistered", "unregister_failed",
                                         "cleanup_failed", "staged_h2d_bytes", "staged_d2h_bytes", "direct_h2d_bytes",
                                         "direct_d2h_bytes", "lines_h2d_bytes", "lines_wait_pack_us", "lines_wait_copy_us", "blocks_retired", "found_page_locked",
                                         "watched_calls", "last_rate_permille")]


def host_stats(reset=False):
    """What the host-buffer paths did since the last reset (``wagg_host_stats_read``): a dict of counters."""
    st = HostStats()
    check(load().wagg_host_stats_read_sized(C.byref(st), C.sizeof(st), 1 if reset else 0), "wagg_host_stats_read_sized")
    return {k: int(getattr(st, k)) for k, _ in HostStats._fields_}


def host_block_plan(T, row_bytes, quantum, n_devices=1):
    """(rows per block, number of blocks) of the row-block pipeline; block i goes to device slot i % n_devices."""
    b, n = C.c_int64(0), C.c_int64(0)
    check(load().wagg_host_block_plan(int(T), int(row_bytes), int(quantum), int(n_devices), C.byref(b), C.byref(n)),
          "wagg_host_block_plan")
    return int(b.value), int(n.value)


class DenseInfo(C.Structure):
    _fields_ = [("G", C.c_int64), ("n_tiles", C.c_int64), ("w_bytes", C.c_int64),
                ("R", C.c_int32), ("n_kt", C.c_int32), ("n_nt", C.c_int32), ("tiled", C.c_int32),
                ("form", C.c_int32), ("elem_bytes", C.c_int32), ("nnz", C.c_int64),
                ("build_s", C.c_double), ("build_upload_s", C.c_double),
                ("est_full_s", C.c_double), ("est_tiles_s", C.c_double), ("est_entries_s", C.c_double), ("walked_entries", C.c_int64),
                ("one_pass_sort", C.c_int32), ("reserved0", C.c_int32)]


class WaggError(RuntimeError):
    pass


class PlanInfo(C.Structure):
    _fields_ = [("nseg_in", C.c_int64), ("nnz", C.c_int64), ("n_groups", C.c_int64),
                ("n_chunks", C.c_int64), ("n_ucells", C.c_int64), ("n_giant", C.c_int64),
                ("n_empty", C.c_int64), ("G", C.c_int64), ("R", C.c_int32), ("lines", C.c_int32),
                ("n_lines128", C.c_int64), ("n_sectors64", C.c_int64), ("n_partial_rows", C.c_int64),
                ("lines_chunks", C.c_int64), ("lines_ucells", C.c_int64), ("lines_lines128", C.c_int64),
                ("n_partial_rows64", C.c_int64), ("lines64_chunks", C.c_int64), ("lines64_ucells", C.c_int64)]


class ApplyDesc(C.Structure):
    """wagg_apply_desc (include/wagg.h): the one descriptor every apply goes through."""
    _fields_ = [("struct_size", C.c_uint64), ("plan_kind", C.c_int32), ("elem", C.c_int32), ("source", C.c_int32), ("transform", C.c_int32),
                ("plan", C.c_void_p), ("x", C.c_void_p), ("x2", C.c_void_p), ("out", C.c_void_p),
                ("T", C.c_int64), ("ldx", C.c_int64), ("ldo", C.c_int64), ("out_pstride", C.c_int64),
                ("layout", C.c_int32), ("out_layout", C.c_int32), ("offset", C.c_double), ("pow_first", C.c_int32), ("n_pow", C.c_int32),
                ("thresholds", C.c_void_p), ("n_thr", C.c_int32), ("flags", C.c_int32), ("ksplit", C.c_int32), ("n_plans", C.c_int32),
                ("devices", C.c_void_p), ("rows", C.c_void_p), ("group", C.c_void_p), ("root", C.c_int32), ("reserved0", C.c_int32),
                ("stream", C.c_void_p)]


STRUCTS = {STRUCT_PLAN_INFO: PlanInfo, STRUCT_DENSE_INFO: DenseInfo, STRUCT_HOST_STATS: HostStats, STRUCT_APPLY_DESC: ApplyDesc}


def _ptr(v):
    """an address for a c_void_p field: ints, c_void_p, ctypes arrays / pointers, None"""
    if v is None:
        return None
    if isinstance(v, int):
        return v
    if isinstance(v, C.c_void_p):
        return v.value
    return C.cast(v, C.c_void_p).value


def run(what, **fields):
    """Fill a wagg_apply_desc and call wagg_apply(): the ONE way this package applies a plan.  Pointer-valued fields take
    addresses, c_void_p or ctypes arrays; the caller keeps the objects behind them alive for the call."""
    d = ApplyDesc()
    d.struct_size = C.sizeof(ApplyDesc)
    d.pow_first = d.n_pow = 1
    for k, v in fields.items():
        setattr(d, k, _ptr(v) if k in _DESC_POINTERS else v)
    check(load().wagg_apply(C.byref(d)), what)


_DESC_POINTERS = frozenset(k for k, t in ApplyDesc._fields_ if t is C.c_void_p)

_lib = None


def load():
    """Load libwagg.so once.  torch is imported first so that one HIP runtime serves both
    (torch bundles libamdhip64.so.7; the loader then reuses it for libwagg.so)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WaggError(
            "HIP engine not built: %s is missing. Run `python -c \"import __graft_entry__ as g; "
            "g.build()\"` or `make -C climate_toolbox_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    import torch  # noqa: F401  (runtime ordering, see docstring)
    L = C.CDLL(LIB_PATH)
    vp, i32p, f64p, f32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_float)
    L.wagg_version.restype = C.c_int
    L.wagg_shard_rows.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.wagg_device_count.restype = C.c_int
    L.wagg_last_error.restype = C.c_char_p
    L.wagg_profile_enable.argtypes = [C.c_int]
    L.wagg_profile_read.argtypes = [f32p, C.c_int, C.POINTER(C.c_int)]
    L.wagg_profile_event_overhead.argtypes = [vp, C.c_int, f32p, f32p]
    u8p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64)
    L.wagg_resolve_cells.argtypes = [f64p, C.c_int64, f64p, C.c_int64, f64p, f64p, C.c_int64, C.c_int, i32p, i64p]
    L.wagg_backup_fill.argtypes = [f64p, f64p, C.c_int64, f64p]
    L.wagg_relabel.argtypes = [f64p, C.c_int64, C.c_double, C.c_double]
    L.wagg_factorize_i64.argtypes = [i64p, u8p, C.c_int64, i32p, i64p, i64p]
    L.wagg_factorize_bytes.argtypes = [C.c_char_p, C.c_int64, u8p, C.c_int64, i32p, i64p, i64p]
    L.wagg_plan_create.argtypes = [i32p, i32p, f64p, C.c_int64, C.c_int64, C.c_int32, C.c_int64,
                                   C.c_int, C.POINTER(vp)]
    L.wagg_plan_destroy.argtypes = [vp]
    L.wagg_plan_get_info.argtypes = [vp, C.POINTER(PlanInfo)]
    L.wagg_plan_get_info_sized.argtypes = [vp, vp, C.c_uint64]
    L.wagg_dense_get_info_sized.argtypes = [vp, vp, C.c_uint64]
    L.wagg_host_stats_read_sized.argtypes = [vp, C.c_uint64, C.c_int]
    L.wagg_struct_size.argtypes = [C.c_int]
    L.wagg_struct_ordinals.argtypes = [C.c_int, vp, C.c_uint64]
    L.wagg_apply.argtypes = [C.POINTER(ApplyDesc)]
    L.wagg_plan_get_den.argtypes = [vp, f64p]
    L.wagg_plan_status.argtypes = [vp, vp]
    for name in ("wagg_apply_f32", "wagg_apply_f64"):
        getattr(L, name).argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int, vp, C.c_int64, C.c_int, vp]
    for name in ("wagg_apply_poly_f32", "wagg_apply_poly_f64"):
        getattr(L, name).argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int, vp,
                                     C.c_int64, C.c_int64, C.c_int, vp]
    for name in ("wagg_apply_poly_host_f32", "wagg_apply_poly_host_f64"):
        getattr(L, name).argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_double, C.c_int, C.c_int, vp, C.c_int64, C.c_int64, C.c_int]
    for name in ("wagg_apply_edd_host_f32", "wagg_apply_edd_host_f64"):
        getattr(L, name).argtypes = [vp, vp, vp, C.c_int64, C.c_int64, C.c_double, f64p, C.c_int, vp, C.c_int64, C.c_int64, C.c_int]
    for name in ("wagg_apply_edd_f32", "wagg_apply_edd_f64"):
        getattr(L, name).argtypes = [vp, vp, vp, C.c_int64, C.c_int64, C.c_int, C.c_double, f64p, C.c_int, vp,
                                     C.c_int64, C.c_int64, C.c_int, vp]
    for name in ("wagg_apply_host_f32", "wagg_apply_host_f64"):
        getattr(L, name).argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int, vp, C.c_int64, C.c_int]
    for name in ("wagg_apply_host_ex_f32", "wagg_apply_host_ex_f64"):
        getattr(L, name).argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int, vp, C.c_int64, C.c_int, C.c_int]
    for name in ("wagg_dense_apply_host_f32", "wagg_dense_apply_host_f64"):
        getattr(L, name).argtypes = [vp, vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int]
    for name in ("wagg_apply_host_multi_f32", "wagg_apply_host_multi_f64", "wagg_dense_apply_host_multi_f32",
                 "wagg_dense_apply_host_multi_f64"):
        getattr(L, name).argtypes = [C.POINTER(vp), i32p, C.c_int, vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int]
    for name in ("wagg_combine_planes_f32", "wagg_combine_planes_f64"):
        getattr(L, name).argtypes = [vp, C.c_int, C.c_int64, f64p, C.c_int64, vp, vp]
    L.wagg_take_axis.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, vp, C.c_int64, vp, vp]
    for name in ("wagg_relayout_f32", "wagg_relayout_f64"):
        getattr(L, name).argtypes = [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp, vp]
    L.wagg_relayout_to_f64.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp, vp]
    L.wagg_upload.argtypes = [vp, vp, C.c_int64]
    L.wagg_host_block_plan.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.wagg_host_stats_read.argtypes = [C.POINTER(HostStats), C.c_int]
    for name in ("wagg_gather_f32", "wagg_gather_f64"):
        getattr(L, name).argtypes = [vp, C.c_int64, C.c_int64, C.c_int, vp, C.c_int64, vp, C.c_int64,
                                     C.c_int, vp]
    for name in ("wagg_transform_poly_f32", "wagg_transform_poly_f64"):
        getattr(L, name).argtypes = [vp, C.c_int64, C.c_double, C.c_int, vp, vp]
    for name in ("wagg_transform_edd_f32", "wagg_transform_edd_f64"):
        getattr(L, name).argtypes = [vp, vp, C.c_int64, C.c_double, f64p, f64p, C.c_int, vp, vp]
    for name in ("wagg_any_less_f32", "wagg_any_less_f64"):
        getattr(L, name).argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int), vp]
    L.wagg_dense_create_synth.argtypes = [C.c_int64, C.c_int32, C.c_uint32, C.POINTER(vp)]
    L.wagg_dense_create_host.argtypes = [f32p, C.c_int64, C.c_int32, C.POINTER(vp)]
    L.wagg_dense_create_from_segments.argtypes = [i32p, i32p, f64p, C.c_int64, C.c_int64, C.c_int32, C.c_int,
                                                  C.POINTER(vp)]
    L.wagg_dense_create_synth_blocklocal.argtypes = [C.c_int64, C.c_int32, C.c_uint32, C.c_double, C.POINTER(vp)]
    L.wagg_dense_create_synth_sparse.argtypes = [C.c_int64, C.c_int32, C.c_uint32, C.c_double, C.POINTER(vp)]
    L.wagg_dense_get_info.argtypes = [vp, C.POINTER(DenseInfo)]
    L.wagg_dense_destroy.argtypes = [vp]
    L.wagg_dense_clone.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.wagg_release_scratch.argtypes = []
    L.wagg_scratch_bytes.argtypes = []
    L.wagg_scratch_bytes.restype = C.c_int64
    L.wagg_dense_get_den.argtypes = [vp, f64p]
    L.wagg_dense_apply_f32.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int, vp]
    L.wagg_dense_apply_poly_f32.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_double, C.c_int, vp, C.c_int64, C.c_int, vp]
    L.wagg_dense_apply_edd_f32.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, C.c_double, C.c_double, vp, C.c_int64,
                                           C.c_int, vp]
    L.wagg_dense_saw_inf.argtypes = [vp, vp, C.POINTER(C.c_int)]
    L.wagg_dense_create_synth_f64.argtypes = [C.c_int64, C.c_int32, C.c_uint32, C.c_double, C.POINTER(vp)]
    L.wagg_dense_create_synth_blocklocal_f64.argtypes = [C.c_int64, C.c_int32, C.c_uint32, C.c_double, C.POINTER(vp)]
    L.wagg_dense_create_host_f64.argtypes = [f64p, C.c_int64, C.c_int32, C.POINTER(vp)]
    L.wagg_dense_create_from_segments_f64.argtypes = [i32p, i32p, f64p, C.c_int64, C.c_int64, C.c_int32, C.c_int, C.POINTER(vp)]
    for name in ("wagg_dense_create_from_csr", "wagg_dense_create_from_csr_f64"):
        getattr(L, name).argtypes = [i64p, i32p, f64p, C.c_int64, C.c_int32, C.c_int, C.POINTER(vp)]
    L.wagg_synth_table_csr.argtypes = [C.c_int64, C.c_int32, C.c_uint32, C.c_double, C.c_int, i64p, i32p, f64p, C.c_int64, i64p]
    L.wagg_dense_apply_f64.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int, vp]
    L.wagg_dense_apply_poly_f64.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_double, C.c_int, vp, C.c_int64, C.c_int, vp]
    L.wagg_dense_apply_edd_f64.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, C.c_double, C.c_double, vp, C.c_int64,
                                           C.c_int, vp]
    L.wagg_synth_field_f32.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, C.c_uint32, C.c_float,
                                       C.c_float, vp]
    L.wagg_synth_field_f64.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, C.c_uint32, C.c_double,
                                       C.c_double, vp]
    L.wagg_shard_group_create.argtypes = [i32p, C.c_int, C.c_int, C.POINTER(vp)]
    L.wagg_shard_group_destroy.argtypes = [vp]
    L.wagg_shard_group_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for name in ("wagg_apply_sharded_f32", "wagg_apply_sharded_f64", "wagg_dense_apply_sharded_f32", "wagg_dense_apply_sharded_f64"):
        getattr(L, name).argtypes = [vp, C.POINTER(vp), C.POINTER(vp), i64p, C.c_int64, vp, C.c_int64, C.c_int]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("wagg_last_error", "wagg_scratch_bytes"):
            fn.restype = C.c_int
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = load().wagg_last_error().decode("utf-8", "replace")
        err = WaggError("%s failed (%d): %s" % (what, rc, msg))
        err.code = int(rc)                 # the wagg_status of the call (include/wagg.h)
        raise err
