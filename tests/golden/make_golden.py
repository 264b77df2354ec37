"""Regenerates tests/golden/*.npz.  Run from the repo root:  python tests/golden/make_golden.py

PROVENANCE (important): the reference cannot be imported in this environment (no xarray/toolz),
so these vectors are NOT outputs of the reference.  Inputs replay the reference's own pytest
fixtures bit for bit (legacy global RandomState stream, tests/test_climate_toolbox.py:33-64 for
lat/lon/clim_data and :86-106 for weights, drawn in that order after np.random.seed(42));
expected outputs come from oracle/ref_numpy.py and are stored only when its three independent
restatements agree to 1e-12 relative.  Known-answer cases carry hand-computed expectations.

reference_held.npz is different: it holds the numeric expectations the reference's OWN tests pin for
the helpers either side of the path, i.e. inputs and expected outputs read off
/root/reference/tests/test_climate_toolbox.py (data, not code):
  :180-187  convert_lons_mono   lon [-156.6, -38.48]  -> [203.4, 321.52]
  :190-197  convert_lons_split  longitude [300, 320]  -> [-60, -40]
  :242-262  snyder_edd  tmin (278.902, 278.23163), tmax (280.4963, 280.7887), threshold 273.15 + 8:
            sum == 0.0 exactly, units "degreedays_281.15K"
  :265-290  snyder_gdd  same fields, thresholds 273.15 + 1 / 273.15 + 8: sum == approx(11, rel 0.1),
            units "degreedays_274.15-281.15K"
These are the only reference-held numbers on or next to the path; the oracle is checked against
them in tests/test_oracle.py and the HIP path in tests/test_gpu_parity.py.
"""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_numpy as O  # noqa: E402
from tests.fixture_replay import reference_fixture, sha256_of  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def agree3(fx, wname, lname):
    args = (fx["temp"], ("lat", "lon", "time"), fx["lat"], fx["lon"], fx["seg_lat"], fx["seg_lon"],
            fx[wname], fx["areawt"], fx[lname])
    a, da, ua = O.agg_scatter(*args, group_dim=lname)
    b, db, ub = O.agg_pandas(*args, group_dim=lname)
    c, dc, uc = O.agg_csr(*args, group_dim=lname)
    assert da == db == dc == (lname, "time"), (da, db, dc)
    assert list(ua) == list(ub) == list(uc)
    for other in (b, c):
        np.testing.assert_allclose(other, a, rtol=1e-12, atol=0)
    return a, np.asarray(ua)


def main():
    fx = reference_fixture()
    assert abs(fx["temp"][0, 0, 0] - 37.454011884736246) < 1e-12
    assert np.isnan(fx["popwt"]).sum() == 20
    out = {k: v for k, v in fx.items() if k != "temp"}
    out["temp_sha256"] = np.array(sha256_of(fx["temp"]))
    for wname in ("popwt", "areawt"):
        for lname in ("ISO", "hierid"):
            val, labels = agree3(fx, wname, lname)
            out["expect_%s_%s" % (wname, lname)] = val
            out["labels_%s" % lname] = labels
    np.savez_compressed(os.path.join(HERE, "reference_fixture.npz"), **out)
    print("reference_fixture.npz: popwt/ISO t=0 ->", out["expect_popwt_ISO"][:, 0])

    # ---- hand-computable known answers (S3-S7, S10) ----------------------------------------
    lat = np.array([10.0, 20.0])
    lon = np.array([100.0, 110.0, 120.0])
    X = np.array([[[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]],
                  [[10.0, 20.0, 30.0], [40.0, 50.0, np.nan]]])   # (time=2, lat=2, lon=3)
    seg = pd.DataFrame({
        "lat":    [10.0, 10.0, 20.0, 20.0, 20.0, 10.0, 20.0, 10.0, 10.0],
        "lon":    [100.0, 110.0, 100.0, 120.0, 120.0, 120.0, 110.0, 110.0, 100.0],
        "areawt": [1.0, 3.0, 2.0, 1.0, 1.0, 5.0, 2.0, 1.0, 4.0],
        "popwt":  [2.0, np.nan, 0.0, -1.0, 3.0, 1.0, 2.0, 6.0, 1.0],
        "lab":    ["b", "b", "a", "a", "a", "c", "z0", "b", None],
    })
    # areawt expectations, by hand:
    #  a: cells (20,100)w2, (20,120)w1, (20,120)w1  -> t0: (4*2+6+6)/4=5 ; t1: (40*2+nan->0+0)/4=20
    #  b: (10,100)w1, (10,110)w3, (10,110)w1 (dup)  -> t0: (1+6+2)/5=1.8 ; t1: (10+60+20)/5=18
    #  c: (10,120)w5                                -> t0: 3 ; t1: 30
    #  z0: (20,110)w2                               -> t0: 5 ; t1: 50
    exp_area = np.array([[5.0, 1.8, 3.0, 5.0], [20.0, 18.0, 30.0, 50.0]])
    # popwt with per-row backup fill (S4): eff = [2, 3(bk), 2(bk), 1(bk), 3, 1, 2, 6, -]
    #  a: (20,100)w2, (20,120)w1, (20,120)w3 -> t0: (8+6+18)/6 = 32/6 ; t1: 80/6
    #  b: (10,100)w2, (10,110)w3, (10,110)w6 -> t0: (2+6+12)/11 = 20/11 ; t1: 200/11
    exp_pop = np.array([[32.0 / 6.0, 20.0 / 11.0, 3.0, 5.0], [80.0 / 6.0, 200.0 / 11.0, 30.0, 50.0]])
    for wname, exp in (("areawt", exp_area), ("popwt", exp_pop)):
        for fn in (O.agg_scatter, O.agg_pandas, O.agg_csr):
            got, dims, labs = fn(X, ("time", "lat", "lon"), lat, lon, seg["lat"].values,
                                 seg["lon"].values, seg[wname].values, seg["areawt"].values,
                                 seg["lab"].values, group_dim="lab")
            assert dims == ("time", "lab") and list(labs) == ["a", "b", "c", "z0"], (dims, labs)
            np.testing.assert_allclose(got, exp, rtol=1e-14)
    np.savez_compressed(
        os.path.join(HERE, "kat_small.npz"), X=X, lat=lat, lon=lon, seg_lat=seg["lat"].values,
        seg_lon=seg["lon"].values, areawt=seg["areawt"].values, popwt=seg["popwt"].values,
        lab=np.array(["b", "b", "a", "a", "a", "c", "z0", "b", ""]), lab_null=np.array(
            [False] * 8 + [True]), expect_areawt=exp_area, expect_popwt=exp_pop,
        labels=np.array(["a", "b", "c", "z0"]))
    print("kat_small.npz written")

    # ---- numbers the reference's own tests hold (tests/test_climate_toolbox.py:177-290) -------
    held = dict(
        mono_lon=np.array([-156.6, -38.48]), mono_expect=np.array([203.4, 321.52]),          # :181-182
        split_lon=np.array([300, 320]), split_expect=np.array([-60, -40]),                   # :191-192
        edd_tmin=np.array([278.902, 278.23163]), edd_tmax=np.array([280.4963, 280.7887]),    # :245,252
        edd_units_in=np.array("K"),
        edd_threshold=np.array(273.15 + 8), edd_sum=np.array(0.0),                           # :257-262
        edd_units=np.array("degreedays_281.15K"),
        gdd_threshold_low=np.array(273.15 + 1), gdd_threshold_high=np.array(273.15 + 8),     # :283-284
        gdd_sum_approx=np.array(11.0), gdd_sum_rel=np.array(0.1),                            # :290
        gdd_units=np.array("degreedays_274.15-281.15K"),
    )
    # the oracle must reproduce every one of them before they are stored
    np.testing.assert_array_equal(O.convert_lons_mono_labels(held["mono_lon"]), held["mono_expect"])
    np.testing.assert_array_equal(O.convert_lons_split(np.zeros(2), ("lon",), held["split_lon"])[1], held["split_expect"])
    assert O.snyder_edd_values(held["edd_tmin"], held["edd_tmax"], float(held["edd_threshold"])).sum() == 0.0
    g = O.snyder_gdd_values(held["edd_tmin"], held["edd_tmax"], float(held["gdd_threshold_low"]),
                            float(held["gdd_threshold_high"])).sum()
    assert abs(g - 11.0) <= 0.1 * 11.0, g
    np.savez_compressed(os.path.join(HERE, "reference_held.npz"), **held)
    print("reference_held.npz written (gdd sum by the oracle = %r)" % g)


if __name__ == "__main__":
    main()
