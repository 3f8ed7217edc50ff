"""Coefficients of gelu_f (ral_device.hpp): 0.5 erfc(a) = 2^-(1 + r(a)), r(a) = a P(a) fitted to -log2 erfc(a) on [0, A] by least squares\nwith Lawson re-weighting, error measured in Phi (weight 0.5 ln2 erfc(a)); prints the fit error, the fp32-evaluated GELU error against\nscipy and the coefficients per degree (degree 8 = the 8 coefficients in the kernel source)."""
import numpy as np
from scipy.special import erfc, erf
np.set_printoptions(precision=17)
A = 5.5
def target(a): return -np.log2(erfc(a))
# Lawson-weighted least squares for r(a) = a * sum_k c_k a^k, error measure: 0.5*ln2*erfc(a)*|dr|  (= |dPhi|)
a = 0.5 * A * (1 - np.cos(np.pi * (np.arange(4000) + 0.5) / 4000))
y = target(a)
for deg in (8, 9, 10):
    w = np.ones_like(a)
    base = 0.5 * np.log(2) * erfc(a)
    V = np.stack([a ** (k + 1) for k in range(deg)], 1)
    for it in range(60):
        W = (w * base)
        c, *_ = np.linalg.lstsq(V * W[:, None], y * W, rcond=None)
        err = np.abs(base * (V @ c - y))
        w = w * (err / err.max() + 1e-3); w /= w.max()
    # fp32 evaluation check on GELU
    x = np.linspace(-8, 8, 2000001)
    c32 = c.astype(np.float32)
    ax = np.minimum(np.abs(x).astype(np.float32) * np.float32(0.70710678118654752), np.float32(A))
    r = np.zeros_like(ax)
    for k in range(deg - 1, -1, -1):
        r = r * ax + c32[k]      # numpy float32 ops (no fma, slightly pessimistic)
    r = r * ax
    h = np.exp2(-(r + np.float32(1.0))).astype(np.float32)          # 0.5 * erfc
    cdf = np.where(x >= 0, np.float32(1.0) - h, h).astype(np.float32)
    g = (x.astype(np.float32) * cdf).astype(np.float64)
    gex = x * 0.5 * erfc(-x / np.sqrt(2))
    # current A-S formula for comparison
    print("deg", deg, "fit max dPhi %.2e" % err.max(), " fp32 GELU max abs err %.2e" % np.abs(g - gex).max(), " at x=%.3f" % x[np.abs(g - gex).argmax()])
    if deg in (8, 9, 10): print("   coeffs:", ", ".join("%.9ef" % v for v in c))
