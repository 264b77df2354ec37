"""Bit-exact replay of the reference's pytest fixtures (inputs only; no reference code runs).

Follows /root/reference/tests/test_climate_toolbox.py:33-64 (lat, lon, clim_data) and :86-106
(weights), drawn from the legacy global RandomState stream in that order after seed(42).
"""
import hashlib

import numpy as np


def sha256_of(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def reference_fixture():
    """Replays fixtures lat, lon, clim_data, weights of tests/test_climate_toolbox.py."""
    lat = np.arange(-89.875, 90, 2)                       # :36
    lon = np.arange(0.125, 360.0, 2)                      # :42
    ntime = 10                                            # :48 (periods=10)
    np.random.seed(42)                                    # :56
    temp = np.random.rand(len(lat), len(lon), ntime) * 100  # :57  dims (lat, lon, time) :60
    lats = np.random.choice(lat, 100)                     # :90
    lons = np.random.choice(lon, 100)                     # :91
    areawt = np.random.random(100)                        # :95
    tmp = np.random.random(100)                           # :96
    tmp[::5] = np.nan                                     # :97
    popwt = tmp                                           # :98
    hierid = np.random.choice(np.arange(1, 25), 100)      # :99
    mapping = {h: np.random.choice(np.arange(1, 5)) for h in hierid}  # :101
    iso = np.array([mapping[i] for i in hierid])          # :103
    return dict(lat=lat, lon=lon, temp=temp, seg_lat=lats, seg_lon=lons, areawt=areawt,
                popwt=popwt, hierid=hierid, ISO=iso)
