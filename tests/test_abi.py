"""The C-ABI library loads and exports every symbol include/wagg.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "wagg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wagg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    from climate_toolbox_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared_symbols()


def test_library_exports_every_declared_symbol():
    from climate_toolbox_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    L = _lib.load()
    for name in _declared_symbols():
        assert hasattr(L, name), name
    assert L.wagg_version() >= 100
    assert L.wagg_device_count() >= 0
    assert isinstance(L.wagg_last_error(), bytes)


def test_bad_arguments_return_codes_not_crashes():
    """Error convention: negative status + message, never an exception across the ABI."""
    from climate_toolbox_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    rc = L.wagg_plan_create(None, None, None, 5, 10, 2, 0, 0, ctypes.byref(h))
    assert rc == -1 and b"NULL" in L.wagg_last_error()
    rc = L.wagg_plan_create(None, None, None, 0, 0, 2, 0, 0, ctypes.byref(h))
    assert rc == -1
    assert L.wagg_plan_get_info(None, None) == -1
    assert L.wagg_dense_apply_f32(None, None, 1, 1, None, 1, 0, None) == -1
    assert L.wagg_plan_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from climate_toolbox_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.WaggError, match="no CPU fallback"):
        _lib.load()


def test_shard_rule_of_the_c_abi_is_the_one_timeshard_uses():
    """wagg_shard_rows (for bindings in other languages) = climate_toolbox_amd.timeshard.shard_bounds."""
    import ctypes as C
    from climate_toolbox_amd import _lib
    from climate_toolbox_amd.timeshard import shard_bounds
    L = _lib.load()
    for T in (0, 1, 7, 365, 10950, 18250):
        for world in (1, 2, 3, 8):
            want = shard_bounds(T, world)
            for rank in range(world):
                a, b = C.c_int64(-1), C.c_int64(-1)
                assert L.wagg_shard_rows(T, world, rank, C.byref(a), C.byref(b)) == 0
                assert (a.value, b.value) == want[rank]
    a, b = C.c_int64(), C.c_int64()
    assert L.wagg_shard_rows(10, 2, 2, C.byref(a), C.byref(b)) == -1
    assert L.wagg_shard_rows(10, 0, 0, C.byref(a), C.byref(b)) == -1


# ---------------------------------------------------------------------------------------------
# structs that cross the boundary by layout (VERDICT r5 Weak #9): sizes, field order, sized getters
# ---------------------------------------------------------------------------------------------
def test_struct_mirrors_have_the_librarys_sizes_and_field_order():
    """Every ctypes mirror in _lib.py has the size the library reports, and the library's ordinal pattern (field k of the C
    struct, in declaration order, holds k) reads 1, 2, 3, ... through the mirror's fields: same order, same types, same
    padding.  Growing a struct on one side only fails here, on the CPU, instead of writing past a buffer on the GPU box."""
    import ctypes as C
    from climate_toolbox_amd import _lib
    L = _lib.load()
    assert L.wagg_struct_size(99) == -1
    for which, S in _lib.STRUCTS.items():
        assert L.wagg_struct_size(which) == C.sizeof(S), S.__name__
        s = S()
        assert L.wagg_struct_ordinals(which, C.byref(s), C.sizeof(s)) == 0
        got = []
        for name, ct in S._fields_:
            v = getattr(s, name)
            got.append(int(v or 0) if ct is C.c_void_p else (int(v) if ct is not C.c_double else v))
        assert got == list(range(1, len(S._fields_) + 1)), (S.__name__, got)
    assert L.wagg_struct_ordinals(99, C.byref(_lib.PlanInfo()), 8) == -1


def test_sized_getters_never_write_past_the_callers_size():
    """A binding built against an OLDER header (smaller struct) gets its prefix and nothing beyond it; one built against a
    NEWER header (larger struct) gets the library's fields and zeros behind them."""
    import ctypes as C
    from climate_toolbox_amd import _lib
    L = _lib.load()
    n = C.sizeof(_lib.HostStats)
    buf = (C.c_uint8 * (n + 64))(*([0xAB] * (n + 64)))
    assert L.wagg_struct_ordinals(_lib.STRUCT_HOST_STATS, buf, 24) == 0           # "old" caller: three int64 fields
    assert list(C.cast(buf, C.POINTER(C.c_int64))[0:3]) == [1, 2, 3] and all(b == 0xAB for b in buf[24:])
    assert L.wagg_host_stats_read_sized(buf, 16, 0) == 0 and all(b == 0xAB for b in buf[24:])
    buf = (C.c_uint8 * (n + 64))(*([0xAB] * (n + 64)))
    assert L.wagg_host_stats_read_sized(buf, n + 64, 0) == 0                      # "new" caller: 64 bytes the library does not know
    assert all(b == 0 for b in buf[n:])
    st = _lib.host_stats()
    assert set(st) == {k for k, _ in _lib.HostStats._fields_} and st["watched_calls"] >= 0


def test_descriptor_entry_point_refuses_what_it_cannot_run():
    """wagg_apply: status codes, never a crash -- NULL / undersized descriptors, unknown enum values, and combinations without
    a kernel path (WAGG_EUNSUPPORTED, the message names the combination; decided before any handle is touched)."""
    import ctypes as C
    from climate_toolbox_amd import _lib
    L = _lib.load()
    assert L.wagg_apply(None) == -1
    d = _lib.ApplyDesc()
    assert L.wagg_apply(C.byref(d)) == -1 and b"struct_size" in L.wagg_last_error()
    d.struct_size = C.sizeof(d)
    assert L.wagg_apply(C.byref(d)) == -1 and b"elem" in L.wagg_last_error()
    d.elem = _lib.T_F32
    assert L.wagg_apply(C.byref(d)) == -1 and b"plan is NULL" in L.wagg_last_error()
    d.plan = 0x1000                                       # never dereferenced by the checks below
    for field, bad in (("plan_kind", 7), ("source", 9), ("transform", 5)):
        setattr(d, field, bad)
        assert L.wagg_apply(C.byref(d)) == -1 and field.encode() in L.wagg_last_error()
        setattr(d, field, 0)
    d.source, d.transform, d.n_plans = _lib.SRC_HOST_MULTI, _lib.XF_POLY, 2
    assert L.wagg_apply(C.byref(d)) == -5 and b"host-multi" in L.wagg_last_error() and b"poly" in L.wagg_last_error()
    d.plan_kind, d.source, d.transform, d.layout = _lib.PLAN_DENSE, _lib.SRC_DEVICE, _lib.XF_NONE, _lib.LAYOUT_GT
    assert L.wagg_apply(C.byref(d)) == -5 and b"dense-family" in L.wagg_last_error()
    d.layout, d.transform, d.n_pow = _lib.LAYOUT_TG, _lib.XF_POLY, 3
    assert L.wagg_apply(C.byref(d)) == -5 and b"one power per call" in L.wagg_last_error()
    with pytest.raises(_lib.WaggError) as e:
        _lib.run("probe", plan=0x1000, plan_kind=_lib.PLAN_DENSE, elem=_lib.T_F64, source=_lib.SRC_SHARDED, transform=_lib.XF_EDD, group=0x1000)
    assert e.value.code == -5
