#!/usr/bin/env python3
"""Fits the epilogue's GELU: erfc(|g| / sqrt 2) ~= exp2(|g| * Q(|g|)), Q a degree-D polynomial, so that
gelu(g) = max(g, 0) - |g|/2 * exp2(|g| Q(|g|)) needs one exponential and no reciprocal.  Weighted least squares on
Chebyshev nodes with Lawson reweighting (approximately minimax in the ABSOLUTE error of gelu)."""
import sys

import numpy as np
from scipy.special import erfc, log_ndtr

D = int(sys.argv[1]) if len(sys.argv) > 1 else 6
HI = 9.0
n = 4000
g = 0.5 * HI * (1 - np.cos(np.pi * (np.arange(n) + 0.5) / n))
g = g[g > 1e-6]
# log2(erfc(g / sqrt2)) = log2(2 * Phi(-g))
target = (np.log(2.0) + log_ndtr(-g)) / np.log(2.0) / g
# d gelu = |g|/2 * erfc * ln2 * g * dQ  -> weight
wbase = 0.5 * g * erfc(g / np.sqrt(2)) * np.log(2.0) * g
V = np.vander(g, D + 1, increasing=True)
w = np.ones_like(g)
for it in range(60):
    ww = wbase * w
    c, *_ = np.linalg.lstsq(V * ww[:, None], target * ww, rcond=None)
    err = np.abs(wbase * (V @ c - target))
    w = w * (err / err.mean()) ** 0.5
    w /= w.mean()
c32 = c.astype(np.float32)
# evaluate in fp32 the way the kernel does
gg = np.linspace(-12, 12, 2000001).astype(np.float32)
ag = np.abs(gg)
q = np.full_like(ag, c32[D])
for k in range(D - 1, -1, -1):
    q = q * ag + c32[k]
e = np.exp2((q * ag).astype(np.float32)).astype(np.float32)
gelu = np.maximum(gg, 0) - (ag * np.float32(0.5)) * e
from scipy.special import erf
ref = 0.5 * gg.astype(np.float64) * (1 + erf(gg.astype(np.float64) / np.sqrt(2)))
d = np.abs(gelu - ref)
print("degree", D, "max abs err of gelu", d.max(), "at g =", gg[d.argmax()])
print("coefficients (c0 .. cD):")
for k, v in enumerate(c32):
    print(f"  c{k} = {v!r}")
