"""Coefficients of the GELU the kernels evaluate (csrc/common.h gelu1 / gelu2):  GELU(x) = h + |h| (1 - exp2(P(min(|x|, 6)))),  h = x / 2,
P(u) = c1 u + ... + c7 u^7 fitted to log2 erfc(u / sqrt 2) on [0, 6] so that the absolute error of GELU, |u| / 2 * |exp2(P(u)) - erfc(u / sqrt 2)|, is minimal
(weighted least squares + Lawson re-weighting).  Prints the coefficients and the errors of the fp32 evaluation (same operation order as the kernels) next to the
Abramowitz & Stegun 7.1.26 form it replaced.      python tools/gelu_fit.py [degree]"""
import sys

import numpy as np
from scipy.special import erf, erfc

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def fit(deg, umax=6.0, n=200001, iters=400):
    u = np.linspace(0, umax, n)
    target = erfc(u / np.sqrt(2))
    y = np.log2(target)
    A = np.vstack([u ** k for k in range(1, deg + 1)]).T
    w = target * np.maximum(u, 0.02)      # d GELU = |u| / 2 * erfc * ln 2 * dP
    lw = np.ones_like(u)
    best = None
    for _ in range(iters):
        W = w * lw
        c, *_ = np.linalg.lstsq(A * W[:, None], y * W, rcond=None)
        err = 0.5 * u * np.abs(np.exp2(A @ c) - target)
        if best is None or err.max() < best[0]:
            best = (err.max(), c.copy())
        lw = lw * (1 + 2 * err / err.max())
        lw /= lw.mean()
    return best


def gelu_new(x, c32, umax=6.0):
    a = np.minimum(np.abs(x), f32(umax)).astype(f32)
    p = np.full_like(a, c32[-1])
    for k in range(len(c32) - 2, -1, -1):
        p = fma(p, a, np.full_like(a, c32[k]))
    p = (p * a).astype(f32)
    r = (f32(1) - np.exp2(p.astype(np.float64)).astype(f32)).astype(f32)
    h = (x * f32(0.5)).astype(f32)
    return fma(np.abs(h), r, h)


def gelu_as(x):
    z = (x * f32(0.70710678118654752440)).astype(f32)
    az = np.abs(z)
    t = (f32(1) / fma(az, np.full_like(az, 0.3275911), np.full_like(az, 1.0))).astype(f32)
    p = fma(t, np.full_like(t, 1.061405429), np.full_like(t, -1.453152027))
    for cst in (1.421413741, -0.284496736, 0.254829592):
        p = fma(p, t, np.full_like(t, cst))
    p = (p * t).astype(f32)
    e = np.exp2(((az * az).astype(f32) * f32(-1.4426950408889634)).astype(np.float64)).astype(f32)
    r = np.copysign((f32(1) - (p * e).astype(f32)).astype(f32), z)
    return ((x * f32(0.5)).astype(f32) * (r + f32(1)).astype(f32)).astype(f32)


if __name__ == "__main__":
    deg = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    e64, c = fit(deg)
    c32 = c.astype(f32)
    x = np.concatenate([-np.linspace(0, 12, 400001)[::-1], np.linspace(0, 12, 400001)]).astype(f32)
    exact = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    print(f"degree {deg}: max |GELU error| with exact arithmetic {e64:.3e}")
    print("coefficients c1..: " + ", ".join(f"{v:.9e}f" for v in c32))
    print(f"fp32 evaluation, x in [-12, 12]: max abs error {np.abs(gelu_new(x, c32).astype(np.float64) - exact).max():.3e}"
          f"   (Abramowitz & Stegun 7.1.26 form in fp32: {np.abs(gelu_as(x).astype(np.float64) - exact).max():.3e})")
