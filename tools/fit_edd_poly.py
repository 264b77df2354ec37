#!/usr/bin/env python3
"""Coefficients of P in csrc/wagg_common.h::snyder_edd1_finite: the fp32 Snyder degree-day band value is
w (sqrt(1 - z^2) - z acos z) / pi = w f(z), and f(x) = (1 - x)^(3/2) P(x) on [0, 1] with a smooth P.  Degree-6 Chebyshev
fit of P, converted to monomials for the Horner form; prints the coefficients (high to low) and the error of the whole
fp32 formula against the fp64 libm evaluation on random fields."""
import numpy as np
from numpy.polynomial import chebyshev as C


def P(x):
    t = 1 - x
    out = np.empty_like(x)
    big = t > 1e-3
    out[big] = (np.sqrt(1 - x[big] ** 2) - x[big] * np.arccos(x[big])) / np.pi / t[big] ** 1.5
    ts = t[~big]      # series at x -> 1 (the closed form cancels there)
    ac = np.sqrt(2 * ts) * (1 + ts / 12 + 3 * ts ** 2 / 160 + 5 * ts ** 3 / 896 + 35 * ts ** 4 / 18432)
    out[~big] = (np.sqrt(2 * ts) * np.sqrt(1 - ts / 2) - (1 - ts) * ac) / np.pi / ts ** 1.5
    return out


xs = (np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) + 1) / 2
c = C.chebfit(2 * xs - 1, P(xs), 6)
xt = np.linspace(0, 1, 200001)[:-1]
print("max |fit - P| on [0, 1):", np.abs(C.chebval(2 * xt - 1, c) - P(xt)).max())
pu = C.cheb2poly(c)
px = np.zeros(1)
for k, a in enumerate(pu):
    term = np.array([1.0])
    for _ in range(k):
        term = np.convolve(term, [-1.0, 2.0])
    px = np.pad(px, (0, max(0, len(term) - len(px))))
    px[:len(term)] += a * term
print("Horner coefficients, high to low:", [repr(float(np.float32(v))) for v in px[::-1]])
rng = np.random.default_rng(0)
n = 2_000_000
f32 = np.float32
tmin = rng.uniform(-20, 40, n).astype(f32)
tmax = (tmin + rng.uniform(0, 20, n).astype(f32)).astype(f32)
e = f32(17.3)
M, w = f32(0.5) * (tmax + tmin), f32(0.5) * (tmax - tmin)
d = M - e
with np.errstate(all="ignore"):
    z = (-d * (f32(1) / w).astype(f32)).astype(f32)
    az = np.fmin(np.abs(z), f32(1))
    t = f32(1) - az
    p = np.full_like(az, f32(px[-1]))
    for a in px[-2::-1]:
        p = (p * az + f32(a)).astype(f32)
    got = (w * ((np.sqrt(t).astype(f32) * t).astype(f32) * p).astype(f32) + np.fmax(d, f32(0))).astype(f32)
    M64, w64 = (tmax.astype(float) + tmin) / 2, (tmax.astype(float) - tmin) / 2
    th = np.arcsin(np.clip((float(e) - M64) / w64, -1, 1))
    ref = np.where(tmin < e, np.where(tmax > e, ((M64 - float(e)) * (np.pi / 2 - th) + w64 * np.cos(th)) / np.pi, 0.0), M64 - float(e))
print("fp32 formula vs fp64 libm: max abs error %.3g, mean %.3g" % (np.abs(got - ref).max(), np.abs(got - ref).mean()))

# fp64: degree-13 fit of P against a 40-digit evaluation (mpmath), for snyder_edd1_finite(double, ...)
try:
    import mpmath as mp
except ImportError:
    mp = None
if mp is not None:
    mp.mp.dps = 40

    def Pm(x):
        x = mp.mpf(x)
        return (mp.sqrt(1 - x * x) - x * mp.acos(x)) / mp.pi / (1 - x) ** mp.mpf(1.5) if x < 1 else 2 * mp.sqrt(2) / (3 * mp.pi)

    nn = 600
    xs = (np.cos(np.pi * (np.arange(nn) + 0.5) / nn) + 1) / 2
    c13 = C.chebfit(2 * xs - 1, np.array([float(Pm(float(x))) for x in xs]), 13)
    xt = np.linspace(0, 1, 20001)
    yt = np.array([float(Pm(float(x))) for x in xt])
    print("fp64, degree 13: max |fit - P| on [0, 1]:", np.abs(C.chebval(2 * xt - 1, c13) - yt).max())
    T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
    for k in range(2, 14):
        a = [mp.mpf(0)] + [2 * v for v in T[k - 1]]
        b = T[k - 2] + [mp.mpf(0)] * (len(a) - len(T[k - 2]))
        T.append([a[i] - b[i] for i in range(len(a))])
    pu = [mp.mpf(0)] * 14
    for k in range(14):
        for i, v in enumerate(T[k]):
            pu[i] += mp.mpf(float(c13[k])) * v
    px13 = [mp.mpf(0)] * 14
    for k, a in enumerate(pu):
        term = [mp.mpf(1)]
        for _ in range(k):
            nt = [mp.mpf(0)] * (len(term) + 1)
            for i, v in enumerate(term):
                nt[i] += -v
                nt[i + 1] += 2 * v
            term = nt
        for i, v in enumerate(term):
            px13[i] += a * v
    print("Horner coefficients, high to low:", [repr(float(v)) for v in px13[::-1]])
