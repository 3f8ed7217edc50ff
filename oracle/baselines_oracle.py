"""CPU restatement of the reference's non-neural comparison baseline (TEST INFRASTRUCTURE ONLY: imported by tests/, never
by the product path).

  wavelet_denoise   local_utils/denoisefunc.py:7-33 - per 1-D record: pywt.wavedec(data, 'db8', level=dwt_max_level),
                    soft threshold of every detail band at 0.04 * max(band), pywt.waverec.

The restatement follows PyWavelets' algorithm (pywt/_extensions/c/convolution.template.c: downsampling_convolution with
MODE_SYMMETRIC, upsampling_convolution_valid_sf; pywt/_multilevel.py: wavedec / waverec; pywt/_thresholding.py: soft) and
the db8 filter bank as PyWavelets tabulates it.  PINNED since round 3: `pywt` is not importable by the image's Python
3.10, but a second interpreter of the build container (/opt/conda/bin/python3.9) has PyWavelets 1.1.1, and
oracle/gen_golden_wavelet.py runs the REFERENCE function there: tests/golden/g8_wavelet.npz holds its outputs for 2-D and
3-D inputs of four lengths, and the pieces (filter bank, dwt_max_level, wavedec bands incl. odd band lengths, waverec,
soft threshold).  tests/test_oracle_golden.py replays them against this file (1e-12); tests/test_baselines_cpu.py keeps
the property checks (spectral factorisation of the Daubechies polynomial, orthonormality, vanishing moments, band
geometry, perfect reconstruction).
"""
import numpy as np

# pywt.Wavelet('db8').dec_lo
DB8_DEC_LO = np.array([
    -0.00011747678412476953, 0.0006754494064505693, -0.00039174037337694705, -0.004870352993451574,
    0.008746094047405777, 0.013981027917398282, -0.044088253930794755, -0.017369301001807547,
    0.12874742662047847, 0.0004724845739132828, -0.2840155429615469, -0.015829105256349306,
    0.5853546836542067, 0.6756307362972898, 0.31287159091429995, 0.05441584224310401])
F = DB8_DEC_LO.size
DB8_DEC_HI = np.array([(-1.0) ** (j + 1) * DB8_DEC_LO[F - 1 - j] for j in range(F)])   # quadrature mirror
DB8_REC_LO = DB8_DEC_LO[::-1].copy()
DB8_REC_HI = DB8_DEC_HI[::-1].copy()


def dwt_max_level(n, filter_len=F):
    """pywt.dwt_max_level: floor(log2(n / (filter_len - 1))), 0 if the record is shorter than the filter."""
    if n < filter_len - 1:
        return 0
    return int(np.floor(np.log2(n / (filter_len - 1.0))))


def _sym(t, n):
    """index into a half-sample symmetric extension  ... x1 x0 | x0 x1 ... x[n-1] | x[n-1] x[n-2] ..."""
    while t < 0 or t >= n:
        t = -t - 1 if t < 0 else 2 * n - 1 - t
    return t


def dwt(x):
    """one level, mode='symmetric': band[o] = sum_j filt[j] * xe[2 o + 1 - j], o < floor((n + F - 1) / 2)"""
    x = np.asarray(x, np.float64)
    n = x.size
    m = (n + F - 1) // 2
    idx = np.array([[_sym(2 * o + 1 - j, n) for j in range(F)] for o in range(m)])
    xe = x[idx]
    return xe @ DB8_DEC_LO, xe @ DB8_DEC_HI


def idwt(a, d):
    """one level: out[2 q + p] = sum_{j < F/2} rec_lo[2 j + p] a[q + F/2 - 1 - j] + rec_hi[2 j + p] d[...], 2 m - F + 2 samples"""
    a = np.asarray(a, np.float64); d = np.asarray(d, np.float64)
    m = a.size
    out = np.zeros(2 * m - F + 2)
    h = F // 2
    for q in range(m - h + 1):
        for j in range(h):
            out[2 * q] += DB8_REC_LO[2 * j] * a[q + h - 1 - j] + DB8_REC_HI[2 * j] * d[q + h - 1 - j]
            out[2 * q + 1] += DB8_REC_LO[2 * j + 1] * a[q + h - 1 - j] + DB8_REC_HI[2 * j + 1] * d[q + h - 1 - j]
    return out


def wavedec(x, level):
    """[cA_level, cD_level, ..., cD_1]"""
    a = np.asarray(x, np.float64)
    det = []
    for _ in range(level):
        a, d = dwt(a)
        det.append(d)
    return [a] + det[::-1]


def waverec(coeffs):
    a = coeffs[0]
    for d in coeffs[1:]:
        if a.size == d.size + 1:      # pywt.waverec drops the surplus sample of an odd-length level
            a = a[:-1]
        a = idwt(a, d)
    return a


def threshold_soft(c, value):
    """pywt.threshold(c, value, 'soft'): c * max(1 - value / |c|, 0)  (zeros stay zero)"""
    mag = np.abs(c)
    with np.errstate(divide="ignore", invalid="ignore"):
        k = np.where(mag > 0, 1.0 - value / mag, 0.0)
    return c * np.clip(k, 0.0, None)


def wavelet_denoise(ecg, threshold=0.04):
    """denoisefunc.py:7-33 on a (rows, L) or (B, leads, L) array"""
    ecg = np.asarray(ecg, np.float64)
    if ecg.ndim == 3:
        return np.stack([wavelet_denoise(r, threshold) for r in ecg])
    out = []
    for row in ecg:
        lev = dwt_max_level(row.size)
        c = wavedec(row, lev)
        for i in range(1, len(c)):
            c[i] = threshold_soft(c[i], threshold * c[i].max())     # (the signed maximum, as the reference writes it)
        out.append(waverec(c))
    return np.array(out)
