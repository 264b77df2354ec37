"""CPU oracle for the weighted grid->region aggregation path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  Nothing under ``climate_toolbox_amd/`` imports it: the product path is the HIP
engine behind ``include/wagg.h`` and fails loudly when that library is missing.

What it restates (reference = /root/reference, read as text; it cannot be imported here because
``xarray`` and ``toolz`` are absent -- an ordinary ModuleNotFoundError):

* ``_reindex_spatial_data_to_regions``      climate_toolbox/aggregations/aggregations.py:8-32
  (live branch :24-27: pointwise ``ds.sel(lon=..., lat=...)`` on a shared ``reshape_index`` dim)
* ``_aggregate_reindexed_data_to_regions``  aggregations.py:35-84
  (:64-66 labels, :69-71 weights, :73 backup fill, :75-82 grouped sums and the division)
* ``weighted_aggregate_grid_to_regions``    aggregations.py:87-124 (:121-122 = the two above)

PARITY STATUS -- "parity unpinned" in the strict sense (no number produced by the reference
exists for this path): the reference's own tests (tests/test_climate_toolbox.py:109-135) pin shapes and
"no NaN" only, never a number, and the reference cannot run here.  The numbers are therefore
pinned by (1) the semantics list S1-S12 of SURVEY.md section 8a, (2) three *independent*
restatements below that must agree to 1e-12 on the reference's own test fixture (legacy-RNG
replay of tests/test_climate_toolbox.py:33-64,86-106) and (3) hand-computable known answers.
"parity pinned by restatement agreement + reference fixtures' shape/no-NaN assertions; no
reference-produced numeric vector exists".

Three restatements (deliberately sharing as little code as possible):
  agg_scatter : dict label lookup -> fancy-index gather -> fp64 multiply -> np.add.at
  agg_pandas  : pandas Index.get_indexer -> DataFrame.groupby(level).sum() (skipna)
  agg_csr     : np.unique/searchsorted lookup -> scipy.sparse CSR  X @ W, den = 1^T W
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------------
# S-switches: the three "best reading" items of SURVEY 8a kept behind named options so that they
# can be flipped if xarray evidence ever appears.
# ----------------------------------------------------------------------------------------------
DROP_NULL_LABELS = True      # S3: rows whose agglev label is NaN/None leave both sums
NAN_DATA_KEEPS_WEIGHT = True  # S6: NaN data adds 0 to the numerator, weight stays in denominator


def effective_weights(w, backup):
    """aggregations.py:73  ``ds[aggwt].where(ds[aggwt] > 0).fillna(weights[backup_aggwt].values)``.

    Per ROW: keep w where w > 0 (NaN > 0 is False), else that row's backup weight (S4)."""
    w = np.asarray(w, dtype=np.float64)
    backup = np.asarray(backup, dtype=np.float64)
    return np.where(w > 0, w, backup)


def _is_null_label(v):
    if v is None:
        return True
    try:
        return bool(np.isnan(v))
    except TypeError:
        return False


def region_codes(labels):
    """Sorted unique labels (xarray groupby order, aggregations.py:78) -> (uniq, code[-1 = null])."""
    labels = np.asarray(labels)
    null = np.array([_is_null_label(v) for v in labels.tolist()], dtype=bool)
    if not DROP_NULL_LABELS and null.any():
        raise ValueError("null labels present and DROP_NULL_LABELS is off")
    good = labels[~null]
    if good.dtype == object:
        uniq = np.array(sorted(set(good.tolist())), dtype=object)
        lut = {v: i for i, v in enumerate(uniq.tolist())}
        cg = np.array([lut[v] for v in good.tolist()], dtype=np.int64)
    else:
        uniq, cg = np.unique(good, return_inverse=True)
    code = np.full(labels.shape[0], -1, dtype=np.int64)
    code[~null] = cg
    return uniq, code


def _lookup_dict(coord, wanted, name):
    """Exact float equality label lookup (S1): ``Dataset.sel`` without ``method=``
    (aggregations.py:27); a label that is absent raises KeyError."""
    lut = {}
    for i, v in enumerate(np.asarray(coord).tolist()):
        lut.setdefault(v, i)
    out = np.empty(len(wanted), dtype=np.int64)
    for j, v in enumerate(np.asarray(wanted).tolist()):
        try:
            out[j] = lut[v]
        except KeyError:
            raise KeyError("%s label %r not found in grid" % (name, v)) from None
    return out


def _split_dims(dims):
    dims = tuple(dims)
    if "lat" not in dims or "lon" not in dims:
        raise KeyError("dims must contain 'lat' and 'lon' (aggregations.py:27), got %r" % (dims,))
    return dims.index("lat"), dims.index("lon")


def out_dims(dims, group_dim):
    """S10: the group dim sits in the slot of the first indexed dim; the other one disappears."""
    ilat, ilon = _split_dims(dims)
    first, second = min(ilat, ilon), max(ilat, ilon)
    res = []
    for i, d in enumerate(dims):
        if i == first:
            res.append(group_dim)
        elif i == second:
            continue
        else:
            res.append(d)
    return tuple(res)


def to_TG(values, dims):
    """Return (X2d[T,G] view/copy in C order with lat-major cells, other_dims, other_shape, nlat, nlon).

    G index = ilat * nlon + ilon.  Pure layout bookkeeping (not part of the reference's maths)."""
    ilat, ilon = _split_dims(dims)
    other = [i for i in range(len(dims)) if i not in (ilat, ilon)]
    v = np.transpose(values, other + [ilat, ilon])
    oshape = v.shape[:-2]
    nlat, nlon = v.shape[-2:]
    T = int(np.prod(oshape)) if oshape else 1
    return v.reshape(T, nlat * nlon), tuple(dims[i] for i in other), oshape, nlat, nlon


def from_TR(out_TR, dims, other_shape, group_dim):
    """Inverse bookkeeping of to_TG for the (T, R) result -> S10 dim order."""
    ilat, ilon = _split_dims(dims)
    first = min(ilat, ilon)
    n_other_before = sum(1 for i in range(len(dims)) if i not in (ilat, ilon) and i < first)
    R = out_TR.shape[1]
    arr = out_TR.reshape(tuple(other_shape) + (R,))
    arr = np.moveaxis(arr, -1, n_other_before)
    return arr, out_dims(dims, group_dim)


# ----------------------------------------------------------------------------------------------
# restatement (a): literal gather -> multiply -> scatter-add -> divide
# ----------------------------------------------------------------------------------------------
def agg_scatter(values, dims, lat, lon, seg_lat, seg_lon, w, backup, labels, group_dim="region"):
    X, _, oshape, nlat, nlon = to_TG(np.asarray(values), dims)
    ilat = _lookup_dict(lat, seg_lat, "lat")
    ilon = _lookup_dict(lon, seg_lon, "lon")
    cell = ilat * nlon + ilon
    w_eff = effective_weights(w, backup)
    uniq, code = region_codes(labels)
    keep = code >= 0
    cell, w_eff, code = cell[keep], w_eff[keep], code[keep]
    gathered = X[:, cell]                                   # aggregations.py:27 (full copy)
    prod = gathered.astype(np.float64) * w_eff[None, :]     # :78, fp64 promotion (S8)
    if NAN_DATA_KEEPS_WEIGHT:
        prod = np.where(np.isnan(prod), 0.0, prod)          # groupby.sum skipna=True (S6)
    R = len(uniq)
    num = np.zeros((X.shape[0], R), dtype=np.float64)
    np.add.at(num.T, code, prod.T)
    den = np.zeros(R, dtype=np.float64)
    np.add.at(den, code, np.where(np.isnan(w_eff), 0.0, w_eff))   # :79, skipna
    with np.errstate(divide="ignore", invalid="ignore"):
        out = num / den[None, :]                            # :77-80, IEEE division (S7)
    arr, od = from_TR(out, dims, oshape, group_dim)
    return arr, od, uniq


# ----------------------------------------------------------------------------------------------
# restatement (b): pandas
# ----------------------------------------------------------------------------------------------
def agg_pandas(values, dims, lat, lon, seg_lat, seg_lon, w, backup, labels, group_dim="region"):
    import pandas as pd
    X, _, oshape, nlat, nlon = to_TG(np.asarray(values), dims)
    ilat = pd.Index(np.asarray(lat)).get_indexer(np.asarray(seg_lat))
    ilon = pd.Index(np.asarray(lon)).get_indexer(np.asarray(seg_lon))
    if (ilat < 0).any() or (ilon < 0).any():
        raise KeyError("segment label not found in grid")
    ws = pd.Series(np.asarray(w, dtype=np.float64))
    w_eff = ws.where(ws > 0).fillna(pd.Series(np.asarray(backup, dtype=np.float64)))
    lab = pd.Series(list(np.asarray(labels).tolist()), dtype=object)
    frame = pd.DataFrame(X[:, ilat * nlon + ilon].astype(np.float64).T * w_eff.values[:, None])
    # groupby drops null keys (dropna=True) and sorts keys; sum skips NaN
    num = frame.groupby(lab.values, sort=True).sum()
    den = w_eff.groupby(lab.values, sort=True).sum()
    uniq = np.array(num.index.tolist(), dtype=object)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = (num.values / den.values[:, None]).T
    arr, od = from_TR(np.ascontiguousarray(out), dims, oshape, group_dim)
    return arr, od, uniq


# ----------------------------------------------------------------------------------------------
# restatement (c): scipy.sparse  (X . W) / (1^T . W)   -- the algebraic form the HIP engine uses
# ----------------------------------------------------------------------------------------------
def agg_csr(values, dims, lat, lon, seg_lat, seg_lon, w, backup, labels, group_dim="region"):
    import scipy.sparse as sp
    X, _, oshape, nlat, nlon = to_TG(np.asarray(values), dims)

    def look(coord, wanted):
        coord = np.asarray(coord)
        order = np.argsort(coord, kind="stable")
        pos = np.searchsorted(coord[order], wanted)
        pos = np.clip(pos, 0, len(coord) - 1)
        idx = order[pos]
        if not np.array_equal(coord[idx], np.asarray(wanted)):
            raise KeyError("segment label not found in grid")
        return idx

    cell = look(lat, seg_lat) * nlon + look(lon, seg_lon)
    w_eff = effective_weights(w, backup)
    uniq, code = region_codes(labels)
    keep = (code >= 0) & ~np.isnan(w_eff)        # NaN weight leaves both sums (skipna on :78/:79)
    W = sp.coo_matrix((w_eff[keep], (cell[keep], code[keep])),
                      shape=(X.shape[1], len(uniq))).tocsr()   # duplicates add (S5)
    Xz = np.where(np.isnan(X), 0.0, X.astype(np.float64))      # S6
    num = np.asarray(Xz @ W) if not sp.issparse(Xz) else None
    den = np.asarray(W.sum(axis=0)).ravel()
    with np.errstate(divide="ignore", invalid="ignore"):
        out = num / den[None, :]
    arr, od = from_TR(out, dims, oshape, group_dim)
    return arr, od, uniq


# ----------------------------------------------------------------------------------------------
# coded form: what crosses the C-ABI (cell_idx, region_code, w_eff already resolved on host)
# ----------------------------------------------------------------------------------------------
def agg_coded(X_TG, cell_idx, region_code, w_eff, R):
    """fp64 oracle on the coded segment table.  X_TG is (T, G); returns (T, R) float64.

    Same arithmetic as agg_scatter after label resolution; used by the GPU parity tests at
    sizes where the label machinery above would dominate."""
    X_TG = np.asarray(X_TG)
    cell_idx = np.asarray(cell_idx, dtype=np.int64)
    region_code = np.asarray(region_code, dtype=np.int64)
    w_eff = np.asarray(w_eff, dtype=np.float64)
    keep = (region_code >= 0) & ~np.isnan(w_eff)
    cell_idx, region_code, w_eff = cell_idx[keep], region_code[keep], w_eff[keep]
    T = X_TG.shape[0]
    num = np.zeros((T, R), dtype=np.float64)
    order = np.argsort(region_code, kind="stable")
    cs, rs, ws = cell_idx[order], region_code[order], w_eff[order]
    den = np.bincount(rs, weights=ws, minlength=R).astype(np.float64)
    bounds = np.searchsorted(rs, np.arange(R + 1))
    # chunk over time to bound the T x nseg temporary
    step = max(1, int(4e7 // max(1, len(cs))))
    for t0 in range(0, T, step):
        xs = X_TG[t0:t0 + step][:, cs].astype(np.float64)
        with np.errstate(invalid="ignore"):
            xs *= ws[None, :]
        xs[np.isnan(xs)] = 0.0                                  # S6
        # segment sums via reduceat on non-empty regions (exact grouping, numpy pairwise order)
        nonempty = bounds[1:] > bounds[:-1]
        if len(cs):
            red = np.add.reduceat(xs, bounds[:-1][nonempty], axis=1)
            num[t0:t0 + step][:, nonempty] = red
    with np.errstate(divide="ignore", invalid="ignore"):
        return num / den[None, :]


def dense_weights_oracle(G, R, seed, fill=1.0):
    """Counter-hash dense W[g,r] in U[0,1) -- must equal wagg_dense_synth on device bit for bit.

    32-bit finaliser (two xorshift-multiply rounds) of (g*R + r) ^ seed*0x9E3779B9, top 24 bits
    scaled by 2^-24 so every value is exactly representable in fp32.  ``fill`` < 1 keeps only
    the entries whose second hash (seed ^ 0x9e3779b9) is below it (wagg_dense_create_synth_sparse:
    c5's uniform-random structure)."""
    g = np.arange(G, dtype=np.uint64)[:, None]
    r = np.arange(R, dtype=np.uint64)[None, :]
    idx = g * np.uint64(R) + r
    W = hash_u01(idx, seed)
    if fill < 1.0:
        keep = hash_u01(idx, np.uint32(seed) ^ np.uint32(0x9e3779b9)) < np.float32(fill)
        W = np.where(keep, W, np.float32(0)).astype(np.float32)
    return W


def hash_u01(idx, seed):
    idx = np.asarray(idx, dtype=np.uint64)
    lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (idx >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        x = lo ^ (hi * np.uint32(0x85EBCA6B)) ^ np.uint32((int(seed) * 0x9E3779B9) & 0xFFFFFFFF)
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def agg_dense(X_TG, W_GR):
    """Dense form (X . W) / (1^T . W) in fp64 with NaN data zeroed (S6)."""
    Xz = np.where(np.isnan(X_TG), 0.0, np.asarray(X_TG, dtype=np.float64))
    W = np.asarray(W_GR, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (Xz @ W) / W.sum(axis=0)[None, :]


# ---------------------------------------------------------------------------------------------
# coordinate standardisation in front of the path
# ---------------------------------------------------------------------------------------------
def convert_lons_split(values, dims, lon):
    """climate_toolbox/utils/utils.py:33-40: relabel ``(lon + 180) % 360 - 180``, then
    ``ds.sel(lon=np.sort(new))`` -- returns (values re-ordered along 'lon', sorted labels)."""
    new = (np.asarray(lon) + 180) % 360 - 180          # integer labels stay integers, like xarray
    order = np.argsort(new, kind="stable")
    return np.take(np.asarray(values), order, axis=tuple(dims).index("lon")), new[order]


def convert_lons_mono_labels(lon):
    """climate_toolbox/utils/utils.py:23-30: relabel ``lon % 360`` and sort ascending."""
    return np.sort(np.asarray(lon) % 360)


def convert_lons_mono(values, dims, lon):
    """utils.py:23-30 on a data array: returns (values re-ordered along 'lon', sorted labels)."""
    new = np.asarray(lon) % 360
    order = np.argsort(new, kind="stable")
    return np.take(np.asarray(values), order, axis=tuple(dims).index("lon")), new[order]


# ---------------------------------------------------------------------------------------------
# grid-level transform in front of the path
# ---------------------------------------------------------------------------------------------
def tas_poly_values(values, power, offset=-273.15):
    """climate_toolbox/transformations/transformations.py:188: ``(ds.tas - 273.15) ** power``,
    evaluated in the data's own dtype (a Python float does not promote a float32 array)."""
    values = np.asarray(values)
    with np.errstate(over="ignore", invalid="ignore"):
        return (values + values.dtype.type(offset)) ** power


def blocklocal_weights_oracle(G, R, seed, fill=0.952, bn=256, bk=32):
    """The c5 "block-local" synthetic weights of wagg_dense_create_synth_blocklocal (include/wagg.h)
    as a dense (G, R) fp32 matrix: run j of 64 cells touches column tile (97 j) mod ceil(R/bn)."""
    n_nt = (R + bn - 1) // bn
    g = np.arange(G, dtype=np.uint64)[:, None]
    r = np.arange(R, dtype=np.uint64)[None, :]
    idx = g * np.uint64(R) + r
    run = (np.arange(G) // bk) // 2
    owner = (97 * run) % n_nt
    inside = (np.arange(R)[None, :] // bn) == owner[:, None]
    keep = hash_u01(idx, np.uint32(seed) ^ np.uint32(0x9e3779b9)) < np.float32(fill)
    return np.where(inside & keep, hash_u01(idx, seed), np.float32(0)).astype(np.float32)


def snyder_edd_values(tasmin, tasmax, threshold):
    """climate_toolbox/transformations/transformations.py:64-87 (Snyder exceedance degree days),
    evaluated in the data's own dtype like xarray would (NumPy scalars of the array's type)."""
    tasmin, tasmax = np.asarray(tasmin), np.asarray(tasmax)
    ft = tasmin.dtype.type
    e, pi = ft(threshold), ft(np.pi)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = (tasmax + tasmin) / ft(2)                                   # :65
        width = (tasmax - tasmin) / ft(2)                                  # :66
        theta = np.arcsin((e - mean) / width)                              # :67
        inner = np.where(tasmax > e, ((mean - e) * (pi / ft(2) - theta) + width * np.cos(theta)) / pi, ft(0))
        return np.where(tasmin < e, inner, mean - e)                       # :73-87


def snyder_gdd_values(tasmin, tasmax, threshold_low, threshold_high):
    """transformations.py:138-140: difference of two EDDs."""
    return snyder_edd_values(tasmin, tasmax, threshold_low) - snyder_edd_values(tasmin, tasmax, threshold_high)
