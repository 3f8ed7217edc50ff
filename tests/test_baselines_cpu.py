"""CPU checks of the wavelet-baseline oracle (oracle/baselines_oracle.py).  pywt is not in this image, so the oracle is
pinned by what defines the published algorithm: the filter bank, the band geometry, perfect reconstruction."""
import math

import numpy as np
import pytest

import baselines_oracle as W


def test_db8_filter_equals_spectral_factorisation_of_the_daubechies_polynomial():
    """Independent derivation: P(y) = sum_{k<8} C(7+k, k) y^k, y = (2 - z - 1/z) / 4; the minimum-phase roots and
    eight zeros at z = -1 give the scaling filter (up to the reversal PyWavelets applies for dec_lo)."""
    N = 8
    c = [math.comb(N - 1 + k, k) for k in range(N)]
    zs = []
    for y in np.roots(c[::-1]):
        r = np.roots([1, -(2 - 4 * y), 1])
        zs.append(r[np.abs(r) < 1][0])
    h = np.poly(zs + [-1] * N).real
    h *= math.sqrt(2) / h.sum()
    assert np.abs(h[::-1] - W.DB8_DEC_LO).max() < 1e-11


def test_db8_bank_is_orthonormal_with_eight_vanishing_moments():
    h, g = W.DB8_DEC_LO, W.DB8_DEC_HI
    assert abs(h.sum() - math.sqrt(2)) < 1e-12
    for k in range(8):
        assert abs(np.dot(h[2 * k:], h[:16 - 2 * k]) - (k == 0)) < 1e-11
        assert abs(np.dot(g[2 * k:], g[:16 - 2 * k]) - (k == 0)) < 1e-11
        assert abs(np.dot(h[2 * k:], g[:16 - 2 * k])) < 1e-11
    t = np.arange(16.0)
    for p in range(8):
        assert abs(np.dot(g, t ** p)) < 1e-3 * 16.0 ** p * 1e-3
    # db2-style sign convention of PyWavelets: dec_hi = [-h15, h14, -h13, ...]; rec filters are the mirror images
    assert g[0] == -h[15] and g[1] == h[14]
    assert np.array_equal(W.DB8_REC_LO, h[::-1]) and np.array_equal(W.DB8_REC_HI, g[::-1])


@pytest.mark.parametrize("n,bands", [(512, [30, 30, 46, 77, 139, 263]), (256, [30, 30, 45, 75, 135]),
                                     (1024, [30, 30, 46, 78, 141, 267, 519]), (30, [22, 22]), (16, [16]), (14, [14])])
def test_band_geometry_and_perfect_reconstruction(n, bands):
    x = np.random.default_rng(n).standard_normal(n)
    lev = W.dwt_max_level(n)
    c = W.wavedec(x, lev)
    assert [b.size for b in c] == bands
    r = W.waverec(c)
    assert r.size == n + (n & 1) and np.abs(r[:n] - x).max() < 1e-10


def test_symmetric_extension_is_half_sample():
    # a constant record has no detail anywhere, the boundary included, and its approximation is sqrt(2) * constant
    a, d = W.dwt(np.full(40, 3.0))
    assert np.abs(d).max() < 1e-11 and np.abs(a - 3.0 * math.sqrt(2)).max() < 1e-11


def test_soft_threshold_and_reference_protocol():
    c = np.array([-3.0, -0.5, 0.0, 0.2, 2.0])
    assert np.allclose(W.threshold_soft(c, 0.5), [-2.5, 0.0, 0.0, 0.0, 1.5])
    x = np.random.default_rng(5).standard_normal((3, 2, 512))
    y = W.wavelet_denoise(x)
    assert y.shape == x.shape
    # the threshold is relative to each band's maximum: the operator is positively homogeneous
    assert np.abs(W.wavelet_denoise(7.5 * x[0]) - 7.5 * y[0]).max() < 1e-9
    # soft thresholding shrinks: the result differs from the input, by less than the input's own size
    assert 0 < np.linalg.norm(y - x) < 0.5 * np.linalg.norm(x)
    # zero threshold = identity
    assert np.abs(W.wavelet_denoise(x[0], threshold=0.0) - x[0]).max() < 1e-10
