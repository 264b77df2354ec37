"""Round 6, GPU: production-shaped masked data at full grid size, and the descriptor entry point against the named symbols."""
import ctypes as C

import numpy as np
import pytest

from tests.test_gpu_parity import RTOL32, RTOL64, _rel_ok

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


@pytest.mark.parametrize("dtype,wname,rtol", [(np.float32, "areawt", RTOL32), (np.float64, "popwt", RTOL64)])
def test_c2_real_with_a_nan_ocean_full_grid(torch_cuda, dtype, wname, rtol):
    """c2-real / c3 tables on the full 720 x 1440 grid with the field the reference's callers actually hold for land-only
    products: every cell the table does not reference is NaN (a masked ocean), 0.5 % of the land cells are NaN too and a few
    hold +-inf.  Every region-timestep of the device apply against the fp64 oracle (S6: NaN data counts 0, the weight stays in
    the denominator), and the host-resident forms -- whole rows, whole lines, quads only -- against the device apply bit for
    bit (the quads-only row carries none of the ocean: the kernel's finite / general decision must not have looked at it)."""
    from climate_toolbox_amd import _lib, engine, synth
    from oracle import ref_numpy as O
    torch = torch_cuda
    lat, lon, df = synth.realistic_segments(string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, wname, "hierid")
    G, R, T = len(lat) * len(lon), len(uniq), 96
    plan = engine.SparsePlan(cell, code, w, G, R, row_len=len(lon))
    rng = np.random.default_rng(23)
    X = (273.15 + 30 * rng.random((T, G))).astype(dtype)
    land = np.zeros(G, dtype=bool)
    land[cell] = True
    X[:, ~land] = np.nan
    lc = np.flatnonzero(land)
    X[rng.integers(0, T, 20000), lc[rng.integers(0, len(lc), 20000)]] = np.nan
    X[rng.integers(0, T, 12), lc[rng.integers(0, len(lc), 12)]] = np.inf
    ref = O.agg_coded(X.astype(np.float64), cell, code, w, R)
    dev = plan.apply(torch.from_numpy(X).cuda()).cpu().numpy()
    fin = np.isfinite(ref)
    assert fin.mean() > 0.99 and np.array_equal(np.isfinite(dev), fin)
    _rel_ok(dev, ref, rtol)
    _lib.host_stats(reset=True)
    for flags in (_lib.HOST_PIN, _lib.HOST_PIN | _lib.HOST_LINES | _lib.HOST_LINES_WHOLE, _lib.HOST_PIN | _lib.HOST_LINES):
        got = plan.apply_host(X, flags=flags)
        assert np.array_equal(got, dev, equal_nan=True), flags
    assert _lib.host_stats()["lines_h2d_bytes"] > 0
    plan.close()


def test_descriptor_and_named_symbols_are_the_same_call(torch_cuda):
    """wagg_apply() with a hand-filled descriptor and the round-1 symbol it replaces give the same bits (the symbol IS a wrapper
    around the descriptor: csrc/wagg_desc.hip) -- segment-table plan, device pointers, both layouts; and a descriptor from an
    "older header" (struct_size cut behind out_pstride) still runs: the fields it lacks read 0 / defaults."""
    from climate_toolbox_amd import _lib, synth
    from climate_toolbox_amd.engine import SparsePlan
    torch = torch_cuda
    L = _lib.load()
    lat, lon, df = synth.realistic_segments(nlat=96, nlon=192, R=150, n_iso=10, seed=3, land_frac=0.2, string_labels=False)
    cell, code, w, uniq = synth.code_segments(df, lat, lon, "areawt", "hierid")
    G, R, T = len(lat) * len(lon), len(uniq), 50
    plan = SparsePlan(cell, code, w, G, R, row_len=len(lon))
    X = torch.from_numpy((280 + 10 * np.random.default_rng(1).standard_normal((T, G))).astype(np.float32)).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for layout, out_layout, Xl in ((0, 0, X), (1, 1, X.t().contiguous())):
        shape = (T, R) if out_layout == 0 else (R, T)
        a = torch.empty(shape, dtype=torch.float32, device="cuda")
        b = torch.empty_like(a)
        rc = L.wagg_apply_f32(plan._h, C.c_void_p(Xl.data_ptr()), T, Xl.shape[1], layout, C.c_void_p(a.data_ptr()), shape[1], out_layout, st)
        assert rc == 0, L.wagg_last_error()
        d = _lib.ApplyDesc(struct_size=C.sizeof(_lib.ApplyDesc), plan_kind=_lib.PLAN_SEGMENT, elem=_lib.T_F32, source=_lib.SRC_DEVICE,
                           plan=plan._h.value, x=Xl.data_ptr(), out=b.data_ptr(), T=T, ldx=Xl.shape[1], ldo=shape[1], layout=layout,
                           out_layout=out_layout, stream=st.value)
        assert L.wagg_apply(C.byref(d)) == 0, L.wagg_last_error()
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        if layout == 0:
            c = torch.empty_like(a)
            d.out = c.data_ptr()
            d.struct_size = _lib.ApplyDesc.layout.offset + 8            # "old header": ends behind `layout` / `out_layout`
            d.stream = 0                                                  # (not part of the old struct: the null stream serves)
            torch.cuda.synchronize()
            assert L.wagg_apply(C.byref(d)) == 0, L.wagg_last_error()
            torch.cuda.synchronize()
            assert torch.equal(a, c)
    plan.close()
