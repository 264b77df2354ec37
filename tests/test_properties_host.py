"""Property tests (hypothesis) of the native label work ahead of the plan -- `wagg_resolve_cells`, `wagg_backup_fill`,
`wagg_factorize_i64` / `_bytes`, `wagg_relabel` (host-only entry points of libwagg.so, no GPU needed) -- against
the oracle's restatements of the reference lines they replace (aggregations.py:27 exact `sel`, :73 backup fill, :78 groupby
order, :144 dateline relabel).  CPU only; the oracle is the checker."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import ref_numpy as O

A = pytest.importorskip("climate_toolbox_amd.aggregations")

_FLOATS = st.floats(allow_nan=False, allow_infinity=False, width=64, min_value=-1e6, max_value=1e6)


@st.composite
def _grid_and_rows(draw):
    nlat, nlon = draw(st.integers(1, 12)), draw(st.integers(1, 12))
    lat = np.array(sorted(draw(st.sets(_FLOATS, min_size=nlat, max_size=nlat))), dtype=np.float64)
    lon = np.array(draw(st.permutations(sorted(draw(st.sets(_FLOATS, min_size=nlon, max_size=nlon))))), dtype=np.float64)
    n = draw(st.integers(0, 60))
    ii = np.array(draw(st.lists(st.integers(0, nlat - 1), min_size=n, max_size=n)), dtype=np.int64)
    jj = np.array(draw(st.lists(st.integers(0, nlon - 1), min_size=n, max_size=n)), dtype=np.int64)
    return lat, lon, ii, jj


@settings(max_examples=80, deadline=None)
@given(_grid_and_rows(), st.booleans())
def test_resolve_cells_is_the_exact_join(case, lon_major):
    """Every row of the table lands on the cell whose labels equal its own bit for bit (S1); the flat index follows the
    layout asked for; unsorted longitude labels (a raw 0-360 file after a relabel) are fine."""
    lat, lon, ii, jj = case
    A._clear_memos() if hasattr(A, "_clear_memos") else None
    cell = A._resolve_cells(lat, lon, lat[ii], lon[jj], lon_major=lon_major)
    want_lat, want_lon = O._lookup_dict(lat, lat[ii], "lat"), O._lookup_dict(lon, lon[jj], "lon")
    want = want_lon * len(lat) + want_lat if lon_major else want_lat * len(lon) + want_lon
    np.testing.assert_array_equal(np.asarray(cell, dtype=np.int64), want)


@settings(max_examples=60, deadline=None)
@given(_grid_and_rows(), st.integers(0, 10**6), st.sampled_from(["lat", "lon"]))
def test_resolve_cells_raises_keyerror_on_any_label_off_the_grid(case, pos, which):
    """One label moved off the grid by one ulp: KeyError, like `Dataset.sel` without `method=` (aggregations.py:27) and
    like the oracle -- never the nearest cell."""
    lat, lon, ii, jj = case
    if len(ii) == 0:
        return
    sa, so = lat[ii].copy(), lon[jj].copy()
    k = pos % len(ii)
    col, grid = (sa, lat) if which == "lat" else (so, lon)
    moved = np.nextafter(col[k], np.inf)
    if moved in grid:                                   # (two grid labels one ulp apart: still on the grid)
        return
    col[k] = moved
    with pytest.raises(KeyError):
        O._lookup_dict(grid, col, which)
    with pytest.raises(KeyError):
        A._resolve_cells(lat, lon, sa, so)


_WEIGHTS = st.one_of(st.floats(allow_nan=True, allow_infinity=True, width=64), st.sampled_from([0.0, -0.0, -1.0, 5e-324]))


@settings(max_examples=100, deadline=None)
@given(st.lists(st.tuples(_WEIGHTS, _WEIGHTS), min_size=0, max_size=50))
def test_backup_fill_is_the_where_of_the_reference(rows):
    """aggregations.py:73: w where w > 0, else that ROW's backup weight -- NaN, zeros of either sign, negatives and
    denormals included; the bits of the value taken are kept (S4)."""
    w = np.array([r[0] for r in rows], dtype=np.float64)
    b = np.array([r[1] for r in rows], dtype=np.float64)
    got, want = A._backup_fill(w, b), O.effective_weights(w, b)
    np.testing.assert_array_equal(got.view(np.int64), want.view(np.int64))


@settings(max_examples=80, deadline=None)
@given(st.lists(st.integers(-2**62, 2**62), min_size=0, max_size=80), st.sampled_from([np.int64, np.int32, np.int16, np.uint8]))
def test_factorize_integers_is_sorted_unique_with_inverse(vals, dtype):
    info = np.iinfo(dtype)
    lab = np.array([info.min + (v - info.min) % (int(info.max) - int(info.min) + 1) for v in vals], dtype=dtype)
    A._TABLE_MEMO.clear() if hasattr(A, "_TABLE_MEMO") and hasattr(A._TABLE_MEMO, "clear") else None
    uniq, codes = A._factorize_labels(lab)
    wu, wc = O.region_codes(lab)
    np.testing.assert_array_equal(np.asarray(uniq), wu)
    np.testing.assert_array_equal(np.asarray(codes, dtype=np.int64), wc)


_LABEL = st.one_of(st.none(), st.text(alphabet=st.characters(min_codepoint=32, max_codepoint=0x2fff, exclude_categories=("Cs",)), max_size=6))


@settings(max_examples=80, deadline=None)
@given(st.lists(_LABEL, min_size=1, max_size=60))
def test_factorize_strings_sorts_like_python_and_drops_nulls(vals):
    """String labels (`hierid`, `ISO`): sorted unique in code-point order like the oracle's sorted(set(...)) -- xarray's
    groupby order (aggregations.py:78) --, None rows coded -1 (S3); the empty string is a label like any other."""
    lab = np.array(vals, dtype=object)
    uniq, codes = A._factorize_labels(lab)
    wu, wc = O.region_codes(lab)
    assert [str(u) for u in uniq] == [str(u) for u in wu]
    np.testing.assert_array_equal(np.asarray(codes, dtype=np.int64), wc)


@settings(max_examples=40, deadline=None)
@given(st.lists(st.sampled_from([180.125, -179.875, 179.875, 0.125, -0.125, 180.12500000000003]), min_size=0, max_size=40))
def test_dateline_relabel_touches_only_the_exact_pixel(vals):
    """aggregations.py:144: `pix_cent_x == 180.125 -> -179.875`, exact equality only."""
    import ctypes as C
    from climate_toolbox_amd import _lib
    L = _lib.load()
    x = np.array(vals, dtype=np.float64)
    want = np.where(x == 180.125, -179.875, x)
    got = x.copy()
    p = got.ctypes.data_as(C.POINTER(C.c_double))
    _lib.check(L.wagg_relabel(p, len(got), 180.125, -179.875), "wagg_relabel")
    np.testing.assert_array_equal(got, want)
