"""Synthetic 2-lead ECG windows with baseline-wander / muscle / electrode-motion noise.

MIT-BIH / NSTDB are not available offline, so throughput and SNR-improvement runs use this generator:
PQRST beats as sums of Gaussians at 360 Hz, heart rate 50-110 bpm with jitter, one R peak placed inside
the central eighth of the window (so the centred R-wave bias of RA-LENet is meaningful), per-window
z-score (reference `np_norm`, local_utils/local_utils.py:261-266), noise scaled to the target input SNR
with the reference formula scale = sqrt(P_sig / 10^(snr/10) / P_noise) (local_utils/local_utils.py:183-189).
Pure numpy; deterministic for a given seed."""
import numpy as np

FS = 360.0
# (offset from R in s, width in s, amplitude) per lead for P, Q, R, S, T
_WAVES = np.array([
    [(-0.20, 0.025, 0.12), (-0.035, 0.010, -0.10), (0.0, 0.011, 1.00), (0.030, 0.012, -0.22), (0.24, 0.045, 0.28)],
    [(-0.21, 0.028, 0.08), (-0.040, 0.011, -0.05), (0.0, 0.013, 0.65), (0.034, 0.014, -0.35), (0.25, 0.050, 0.18)],
])


def _beats(rng, n, L, leads):
    t = np.arange(L) / FS
    x = np.zeros((n, leads, L))
    for i in range(n):
        rr = 60.0 / rng.uniform(50, 110)
        r0 = (L / 2 + rng.uniform(-L / 16, L / 16)) / FS
        k0 = int(np.floor((0 - r0) / rr)) - 1
        k1 = int(np.ceil((L / FS - r0) / rr)) + 1
        amp = rng.uniform(0.8, 1.2)
        for k in range(k0, k1 + 1):
            rk = r0 + k * rr * (1 + (0 if k == 0 else rng.normal(0, 0.03)))
            for ld in range(leads):
                for (off, wid, a) in _WAVES[ld % 2]:
                    x[i, ld] += amp * a * np.exp(-0.5 * ((t - rk - off * np.sqrt(rr / 0.8)) / wid) ** 2)
    return x


def _band_noise(rng, shape, f_lo, f_hi):
    n = shape[-1]
    spec = np.fft.rfft(rng.standard_normal(shape), axis=-1)
    f = np.fft.rfftfreq(n, 1 / FS)
    spec *= ((f >= f_lo) & (f <= f_hi))
    return np.fft.irfft(spec, n, axis=-1)


def noise(rng, kind, shape):
    """bw: 0.05-0.7 Hz drift; ma: 5-50 Hz bursts; em: steps and spikes; emb: their sum."""
    n, leads, L = shape
    if kind == "bw":
        # drift needs a long support: synthesise 8x the window and cut
        z = _band_noise(rng, (n, leads, 8 * L), 0.05, 0.7)
        return z[..., 3 * L:4 * L]
    if kind == "ma":
        z = _band_noise(rng, shape, 5.0, 50.0)
        env = np.ones(shape)
        for i in range(n):
            c, w = rng.uniform(0, L), rng.uniform(L / 8, L / 2)
            env[i] = 0.3 + np.exp(-0.5 * ((np.arange(L) - c) / w) ** 2)
        return z * env
    if kind == "em":
        z = np.zeros(shape)
        for i in range(n):
            for ld in range(leads):
                for _ in range(rng.integers(1, 4)):
                    p = rng.integers(0, L)
                    z[i, ld, p:] += rng.normal(0, 1.0)
                for _ in range(rng.integers(0, 3)):
                    p = rng.integers(0, L)
                    z[i, ld] += rng.normal(0, 3.0) * np.exp(-0.5 * ((np.arange(L) - p) / 2.0) ** 2)
        return z - z.mean(-1, keepdims=True)
    if kind == "emb":
        out = 0
        for k in ("bw", "ma", "em"):
            z = noise(rng, k, shape)
            out = out + z / np.sqrt((z ** 2).mean((1, 2), keepdims=True) + 1e-12)
        return out
    raise ValueError(kind)


def make_dataset(n=10000, leads=2, L=256, noise_name="emb", snr_db=0.0, seed=2023):
    """-> (noisy, clean) float32 arrays of shape (n, leads, L); the on-disk counterpart is
    data/dict_data/{m4,m2,0,p2,p4}/{bw,ma,em,emb}.npy + data/dict_data/ecg.npy (data_utils.py:92-117)."""
    rng = np.random.default_rng(seed)
    clean = _beats(rng, n, L, leads)
    clean = clean - clean.mean(-1, keepdims=True)
    clean = clean / clean.std(-1, keepdims=True)                      # np_norm over the window
    z = noise(rng, noise_name, (n, leads, L))
    p_sig = (clean ** 2).mean((1, 2), keepdims=True)
    p_noise = (z ** 2).mean((1, 2), keepdims=True)
    scale = np.sqrt(p_sig / (10 ** (snr_db / 10)) / p_noise)           # single_snr_noise_add
    return (clean + scale * z).astype(np.float32), clean.astype(np.float32)


def split_8000_2000(noisy, clean, seed=2023):
    """main.py:52-58 protocol: 80/20 split of the selected windows (own RNG, documented deviation)."""
    idx = np.random.default_rng(seed).permutation(len(noisy))
    k = int(0.8 * len(idx))
    return (noisy[idx[:k]], clean[idx[:k]]), (noisy[idx[k:]], clean[idx[k:]])
