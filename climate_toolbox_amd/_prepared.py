"""A weights table coded ONCE: PreparedWeights / prepare_weights.

Split out of aggregations.py in round 6; aggregations.py re-exports both names."""
from __future__ import annotations

import threading

import numpy as np
import pandas as pd

from ._labels import _backup_fill, _factorize_labels, _resolve_cells
from ._memo import _f64, _fingerprint, _frozen


# ----------------------------------------------------------------------------------------------
# A weights table coded ONCE: a pipeline that loops over variables, files or years with one table pays
# the label join, the backup fill, the factorisation and the fingerprints a single time instead of re-hashing ~16 MB of
# table columns on every call to find out that nothing changed.
# ----------------------------------------------------------------------------------------------
class PreparedWeights:
    """A SNAPSHOT of one segment-weights table, coded for one ``(aggwt, agglev, backup_aggwt)``: backup-filled weights
    (aggregations.py:73), sorted unique region labels and per-row codes (:78), and -- per grid it has met -- the resolved
    cell of every row (:27) and the key of its plan.  Pass it as ``weights`` to :func:`weighted_aggregate_grid_to_regions`
    (or the two helpers) wherever the DataFrame went; ``aggwt`` / ``agglev`` of the call must be the ones it was prepared
    for.  It copies what it needs: later edits of the DataFrame do not reach it (a bare DataFrame is still fingerprinted by
    content on every call, so edits of THAT are always seen)."""

    def __init__(self, df, aggwt, agglev, backup_aggwt="areawt"):
        self.aggwt, self.agglev, self.backup_aggwt = aggwt, agglev, backup_aggwt
        self.seg_lat = _frozen(np.array(df["lat"].values, dtype=np.float64, copy=True))
        self.seg_lon = _frozen(np.array(df["lon"].values, dtype=np.float64, copy=True))
        self.w_eff = _frozen(_backup_fill(df[aggwt].values, df[backup_aggwt].values))
        self.labels = _frozen(np.array(df[agglev].values, copy=True))
        uniq, codes = _factorize_labels(self.labels)
        self.uniq, self.codes = uniq, codes                       # (codes is read-only; uniq is handed out as a copy)
        self.nseg = len(self.w_eff)
        self._grids = {}        # (nlat, nlon, lat[0], lon[0]) -> [(lat, lon, cell, ilat, ilon)]
        self._plan_keys = {}    # (id of a cell array this object owns, G, R, row_len, is_f32, layout) -> plan key
        self._lock = threading.Lock()

    # DataFrame-like access for code that reads the columns the reference reads (weights[aggwt].values ...)
    def __getitem__(self, col):
        if col == "lat":
            return pd.Series(self.seg_lat)
        if col == "lon":
            return pd.Series(self.seg_lon)
        if col == self.agglev:
            return pd.Series(self.labels)
        if col == self.aggwt:
            return pd.Series(self.w_eff)
        raise KeyError("%r: this PreparedWeights holds lat, lon, %r (backup-filled) and %r" % (col, self.aggwt, self.agglev))

    def __len__(self):
        return self.nseg

    def check(self, aggwt, agglev, backup_aggwt="areawt"):
        if (aggwt, agglev, backup_aggwt) != (self.aggwt, self.agglev, self.backup_aggwt):
            raise ValueError("weights were prepared for aggwt=%r, agglev=%r, backup_aggwt=%r; the call asks for %r, %r, %r"
                             % (self.aggwt, self.agglev, self.backup_aggwt, aggwt, agglev, backup_aggwt))

    def cells_for(self, lat, lon):
        """(cell, ilat, ilon) of every row on this grid (exact label match, KeyError on a miss: S1), resolved once per grid."""
        lat, lon = np.asarray(lat), np.asarray(lon)
        gk = (len(lat), len(lon), float(lat[0]) if len(lat) else 0.0, float(lon[0]) if len(lon) else 0.0)
        with self._lock:
            for glat, glon, cell, ilat, ilon in self._grids.get(gk, ()):
                if np.array_equal(glat, lat) and np.array_equal(glon, lon):
                    return cell, ilat, ilon
        cell = _resolve_cells(lat, lon, self.seg_lat, self.seg_lon)
        ilat, ilon = _frozen((cell // len(lon)).astype(np.int64)), _frozen((cell % len(lon)).astype(np.int64))
        with self._lock:
            self._grids.setdefault(gk, []).append((_f64(lat).copy(), _f64(lon).copy(), cell, ilat, ilon))
        return cell, ilat, ilon

    def plan_key(self, cell_idx, G, R, row_len, is_f32, layout):
        """Key of the plan of (this table, this cell index): hashed once per cell array this object owns, else per call."""
        extra = repr((int(G), int(R), int(row_len), bool(is_f32), layout))
        with self._lock:
            owned = any(cell_idx is c for entries in self._grids.values() for _, _, c, _, _ in entries)
        if not owned:                                  # e.g. a lon-permuted or lon-major index: a fresh array every call
            return _fingerprint(cell_idx, self.codes, self.w_eff, extra=extra)[0]
        k = (id(cell_idx), extra)
        with self._lock:
            key = self._plan_keys.get(k)
        if key is None:                                # the key a bare DataFrame of the same content gets: one plan serves both
            key = _fingerprint(cell_idx, self.codes, self.w_eff, extra=extra)[0]
            with self._lock:
                self._plan_keys[k] = key
        return key


def prepare_weights(weights, aggwt, agglev, backup_aggwt="areawt", lat=None, lon=None):
    """Code a segment-weights table (DataFrame, or the path of its CSV) once for ``(aggwt, agglev)``; with the grid's
    ``lat`` / ``lon`` labels the cells are resolved now (KeyError on a label that is not on the grid), else at first use."""
    if isinstance(weights, PreparedWeights):
        weights.check(aggwt, agglev, backup_aggwt)
        prep = weights
    else:
        if isinstance(weights, str):
            from .aggregations import prepare_spatial_weights_data
            weights = prepare_spatial_weights_data(weights)
        prep = PreparedWeights(weights, aggwt, agglev, backup_aggwt)
    if lat is not None and lon is not None:
        prep.cells_for(lat, lon)
    return prep
