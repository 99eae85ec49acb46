"""Fit the degree-6 polynomial of log2 Phi(-a) on [0, 6] that `gelu_f` (csrc/kernels_common.hip.h) evaluates.

gelu(x) = max(x,0) - |x| Phi(-|x|); the subtracted term is |x| * exp2(h(min(|x|, 6))).  The fit minimises (iteratively
re-weighted least squares towards the minimax solution) the absolute error of that term, i.e. weight a*Phi(-a) on h.
Prints the coefficients and the max abs error of an fp32 Horner evaluation against an fp64 GELU over [-12, 12].
"""
import numpy as np
from scipy.special import erfc, log_ndtr

DEG, A = 6, 6.0


def fit(deg=DEG, hi=A, iters=400):
    a = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) * hi / 2 + hi / 2
    h = log_ndtr(-a) / np.log(2)
    w = a * np.exp(log_ndtr(-a)) + 1e-5
    ww = w.copy()
    for _ in range(iters):
        c = np.polynomial.polynomial.polyfit(a, h, deg, w=ww)
        err = np.abs(np.polynomial.polynomial.polyval(a, c) - h) * w
        ww = ww * (1 + 2 * err / err.max()) / 2
    return c


def gelu_f32(x, c):
    c32 = c.astype(np.float32)
    a = np.minimum(np.abs(x), np.float32(A))
    p = np.full_like(a, c32[-1])
    for k in range(len(c32) - 2, -1, -1):
        p = p * a + c32[k]
    return np.maximum(x, 0) - np.abs(x) * np.exp2(p).astype(np.float32)


if __name__ == "__main__":
    c = fit()
    print("coefficients (a^0 .. a^%d):" % DEG, ", ".join("%.10e" % v for v in c))
    x = np.linspace(-12, 12, 4000001).astype(np.float32)
    xd = x.astype(np.float64)
    ref = xd * 0.5 * erfc(-xd / np.sqrt(2))
    e = np.abs(gelu_f32(x, c) - ref)
    print("max abs error of the fp32 evaluation: %.3e at x = %.4f" % (e.max(), x[e.argmax()]))
