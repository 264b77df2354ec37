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
