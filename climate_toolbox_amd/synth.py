"""Synthetic workloads of BASELINE.json / SURVEY.md section 8d (host-side generators, fixed seeds).

These build INPUTS only (grids, fields, segment-weight tables shaped like the reference's
``prepare_spatial_weights_data`` output, aggregations.py:141-150); no aggregation arithmetic
lives here.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

N_IMPACT_REGIONS = 24378     # transformations.py:151-153 (hierid count)


def grid_2deg():
    """The reference's test grid (tests/test_climate_toolbox.py:36,42)."""
    return np.arange(-89.875, 90, 2), np.arange(0.125, 360.0, 2)


def grid_quarter_deg():
    """0.25-degree production grid, pixel centres on x.125/x.375/... (aggregations.py:144)."""
    return -89.875 + 0.25 * np.arange(720), -179.875 + 0.25 * np.arange(1440)


def c1_workload(T=365, seed=0):
    """c1: 2-degree grid, 100 block regions, 5 % of cells split across two regions, fp64."""
    lat, lon = grid_2deg()
    rng = np.random.default_rng(seed)
    coslat = np.cos(np.deg2rad(lat))
    tas = 273.15 + 30.0 * coslat[None, :, None] + 10.0 * rng.standard_normal((T, len(lat), len(lon)))
    ii, jj = np.meshgrid(np.arange(len(lat)), np.arange(len(lon)), indexing="ij")
    region = (ii // 9) * 10 + (jj // 18)
    rows = dict(lat=lat[ii.ravel()], lon=lon[jj.ravel()], region=region.ravel().copy(),
                frac=np.ones(ii.size))
    split = rng.random(ii.size) < 0.05
    other = (rows["region"][split] + 1) % 100
    f = rng.uniform(0.2, 0.8, split.sum())
    rows["frac"][split] = f
    df = pd.DataFrame({
        "lat": np.concatenate([rows["lat"], rows["lat"][split]]),
        "lon": np.concatenate([rows["lon"], rows["lon"][split]]),
        "hierid": np.concatenate([rows["region"], other]),
        "frac": np.concatenate([rows["frac"], 1.0 - f]),
    })
    df["areawt"] = np.cos(np.deg2rad(df["lat"].values)) * rng.uniform(0.5, 1.0, len(df)) * df["frac"].values
    df["popwt"] = rng.lognormal(0.0, 2.0, len(df))
    df.loc[rng.random(len(df)) < 0.2, "popwt"] = np.nan
    df["ISO"] = df["hierid"] // 10
    df.index.names = ["reshape_index"]
    return lat, lon, tas, df.drop(columns=["frac"])


def _smooth_noise(nlat, nlon, rng, coarse=(36, 72)):
    from scipy.ndimage import zoom
    z = rng.standard_normal(coarse)
    z = np.concatenate([z, z[:, :1]], axis=1)
    f = zoom(z, (nlat / coarse[0], nlon / coarse[1] * coarse[1] / (coarse[1] + 1)), order=3)
    return f[:nlat, :nlon]


def realistic_segments(nlat=720, nlon=1440, R=N_IMPACT_REGIONS, land_frac=0.30, n_iso=200, seed=2,
                       lat=None, lon=None, string_labels=True):
    """c2-real / c3 segment table: Voronoi-like compact regions over ~30 % "land" cells, border
    cells split between two (sometimes three) regions -> 1-3 segments per land cell.

    Returns a DataFrame with the columns of the reference's weights file after
    prepare_spatial_weights_data: lat, lon, areawt, popwt, hierid, ISO (index reshape_index)."""
    from scipy.spatial import cKDTree
    if lat is None or lon is None:
        lat, lon = grid_quarter_deg() if (nlat, nlon) == (720, 1440) else (
            -90 + (np.arange(nlat) + 0.5) * 180.0 / nlat, -180 + (np.arange(nlon) + 0.5) * 360.0 / nlon)
    rng = np.random.default_rng(seed)
    noise = _smooth_noise(nlat, nlon, rng)
    land = noise > np.quantile(noise, 1.0 - land_frac)
    li, lj = np.nonzero(land)
    nland = len(li)
    R = min(R, nland)
    seeds = rng.choice(nland, size=R, replace=False)
    pts = np.stack([li, lj], axis=1).astype(np.float64)
    _, owner = cKDTree(pts[seeds]).query(pts, k=1)
    # countries: coarse Voronoi over the region seeds
    iso_seeds = rng.choice(R, size=min(n_iso, R), replace=False)
    _, iso_of_region = cKDTree(pts[seeds][iso_seeds]).query(pts[seeds], k=1)
    # border cells: a 4-neighbour owned by another region donates a second (third) segment
    own_grid = np.full((nlat, nlon), -1, dtype=np.int64)
    own_grid[li, lj] = owner
    seg_i, seg_j, seg_r, seg_f = [li], [lj], [owner], [np.ones(nland)]
    for di, dj, p in ((0, 1, 0.6), (1, 0, 0.4)):
        ni, nj = np.clip(li + di, 0, nlat - 1), (lj + dj) % nlon
        nb = own_grid[ni, nj]
        m = (nb >= 0) & (nb != owner) & (rng.random(nland) < p)
        f = rng.uniform(0.1, 0.5, m.sum())
        seg_f[0][m] -= f * seg_f[0][m]
        seg_i.append(li[m]); seg_j.append(lj[m]); seg_r.append(nb[m]); seg_f.append(f)
    si, sj = np.concatenate(seg_i), np.concatenate(seg_j)
    sr, sf = np.concatenate(seg_r), np.concatenate(seg_f)
    perm = rng.permutation(len(si))                      # table row order is arbitrary (S5/S12)
    si, sj, sr, sf = si[perm], sj[perm], sr[perm], sf[perm]
    areawt = np.cos(np.deg2rad(lat[si])) * sf
    popwt = rng.lognormal(0.0, 2.0, len(si))
    kill = rng.random(len(si))
    popwt[kill < 0.1] = np.nan
    popwt[(kill >= 0.1) & (kill < 0.2)] = 0.0
    iso = iso_of_region[sr]
    if string_labels:
        iso_lab = np.array(["C%03d" % k for k in range(iso.max() + 1)], dtype=object)[iso]
        hier = np.array(["%s.R%05d" % (a, b) for a, b in zip(iso_lab, sr)], dtype=object)
    else:
        iso_lab, hier = iso, sr
    df = pd.DataFrame({"lat": lat[si], "lon": lon[sj], "areawt": areawt, "popwt": popwt,
                       "hierid": hier, "ISO": iso_lab})
    df.index.names = ["reshape_index"]
    return lat, lon, df


def code_segments(df, lat, lon, aggwt, agglev, backup="areawt"):
    """Host-side label resolution for callers that drive the C-ABI directly (bench, tests):
    exact label lookup (S1), backup fill (S4), sorted-unique factorisation (S3)."""
    from .aggregations import _backup_fill, _factorize_labels, _resolve_cells
    cell = _resolve_cells(lat, lon, df["lat"].values, df["lon"].values)
    w_eff = _backup_fill(df[aggwt].values, df[backup].values)
    uniq, codes = _factorize_labels(df[agglev].values)
    return cell, codes, w_eff, uniq
