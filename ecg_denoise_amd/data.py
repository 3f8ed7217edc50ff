"""Data path in the reference's on-disk format.

`data/dict_data/{m4,m2,0,p2,p4}/{bw,ma,em,emb}.npy` hold the noisy windows (N, 2, L) at input SNR
-4/-2/0/+2/+4 dB and `data/dict_data/ecg.npy` the clean ones; index i pairs them
(reference local_utils/data_utils.py:88-117).  `EcgDataset` reads exactly that layout, `make_loaders`
reproduces main.py:52-60 (10 000 random windows, 80/20 split, shuffled batches of FloatTensors), and
`write_synthetic_dataset` fills the layout from ecg_denoise_amd/synth.py when MIT-BIH/NSTDB are absent."""
import os
import random

import numpy as np

NOISE_INDEX = ['m4', 'm2', '0', 'p2', 'p4']
TRUE_NOISE = [-4, -2, 0, 2, 4]
NOISE_TYPES = ['bw', 'ma', 'em', 'emb']


class EcgDataset:
    def __init__(self, noise_name='bw', noise_intensity=0, path=None):
        if isinstance(noise_name, str):
            noise_name = [noise_name]
        if noise_intensity not in TRUE_NOISE:
            raise ValueError("noise intensity should be in [-4, -2, 0, 2, 4]")
        if path is None:
            path = "./data/dict_data/" if os.path.exists("./data/dict_data/") else "../data/dict_data/"
        sub = os.path.join(path, NOISE_INDEX[TRUE_NOISE.index(noise_intensity)])
        self.data = np.concatenate([np.load(os.path.join(sub, n + '.npy')) for n in noise_name], axis=0)
        self.ground_data = np.load(os.path.join(path, 'ecg.npy'))

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, index):
        return self.data[index], self.ground_data[index]


class _Loader:
    """Shuffled mini-batches of (noisy, clean) float32 arrays: the custom_collate_fn of main.py:45-48
    without the per-sample Python tuples."""

    def __init__(self, noisy, clean, batch_size, shuffle, seed, drop_last=False):
        self.noisy, self.clean, self.bs, self.shuffle = noisy, clean, batch_size, shuffle
        self.rng = np.random.default_rng(seed)
        self.drop_last = drop_last
        self.dataset = noisy

    def __len__(self):
        n = len(self.noisy)
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def __iter__(self):
        idx = self.rng.permutation(len(self.noisy)) if self.shuffle else np.arange(len(self.noisy))
        for i in range(len(self)):
            j = idx[i * self.bs:(i + 1) * self.bs]
            yield self.noisy[j], self.clean[j]


def make_loaders(dataset, batch_size=32, n_select=10000, train_ratio=0.8, seed=2023):
    """main.py:52-60: random.sample(n_select) -> 80/20 split -> two shuffled loaders."""
    rnd = random.Random(seed)
    sel = rnd.sample(range(len(dataset)), min(n_select, len(dataset)))
    rnd.shuffle(sel)
    k = int(train_ratio * len(sel))
    tr, te = np.array(sel[:k]), np.array(sel[k:])
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return (_Loader(f32(dataset.data[tr]), f32(dataset.ground_data[tr]), batch_size, True, seed),
            _Loader(f32(dataset.data[te]), f32(dataset.ground_data[te]), batch_size, True, seed + 1))


def write_synthetic_dataset(path, n=10000, leads=2, L=256, seed=2023, noise_types=NOISE_TYPES, intensities=TRUE_NOISE):
    from . import synth
    os.makedirs(path, exist_ok=True)
    clean = None
    for snr in intensities:
        sub = os.path.join(path, NOISE_INDEX[TRUE_NOISE.index(snr)])
        os.makedirs(sub, exist_ok=True)
        for nt in noise_types:
            noisy, c = synth.make_dataset(n, leads, L, nt, float(snr), seed)   # same seed -> same clean windows
            clean = c if clean is None else clean
            np.save(os.path.join(sub, nt + '.npy'), noisy)
    np.save(os.path.join(path, 'ecg.npy'), clean)
    return path


# ---------------------------------------------------------------------------------------------------------------
# Record iterators: the step in front of the model (reference local_utils/local_utils.py:116-130,
# `batch_norm_snr_iter`), with the z-score, the SNR-scaled noise add and the windowing done on the GPU
# (`ral_prep_windows`) instead of numpy + torch.FloatTensor + einops on the host.
# ---------------------------------------------------------------------------------------------------------------
def prep_windows(signal, noise, snr, L=256, stream=None):
    """One `batch_norm_snr_iter` iteration after the file reads: `signal` and `noise` are (T, leads) device tensors
    (ADC units; any float/int dtype), T a multiple of L.  Returns fp32 device tensors (noisy, clean), each
    (T // L, leads, L): `clean` is the per-lead z-score of the segment (`np_norm(..., dim=0)`), `noisy` adds the noise
    scaled to `snr` dB (`Gnoisegen`)."""
    import torch
    from . import _lib
    if not signal.is_cuda or not noise.is_cuda:
        raise _lib.RalError("prep_windows runs on the GPU: pass device tensors (there is no CPU fallback)")
    sig = signal.to(torch.float32).contiguous()
    noi = noise.to(torch.float32).contiguous()
    if sig.dim() != 2 or sig.shape != noi.shape:
        raise _lib.RalError(f"signal and noise must both be (T, leads); got {tuple(sig.shape)} and {tuple(noi.shape)}")
    T, leads = sig.shape
    if T % L:
        raise _lib.RalError(f"segment length {T} is not a multiple of the window length {L}")
    noisy = torch.empty(T // L, leads, L, dtype=torch.float32, device=sig.device)
    clean = torch.empty_like(noisy)
    sums = torch.empty(2 * leads + 1, dtype=torch.float64, device=sig.device)
    cur = torch.cuda.current_stream(sig.device)
    s = stream if stream is not None else cur
    if s != cur:
        s.wait_stream(cur)                     # the float32 copies above were made on the current stream
    _lib.check(_lib.lib().ral_prep_windows(sig.data_ptr(), noi.data_ptr(), T, leads, L, float(snr), sums.data_ptr(),
                                           noisy.data_ptr(), clean.data_ptr(), s.cuda_stream))
    if s != cur:                               # the caching allocator must not hand these out while `s` still uses them
        for t in (sig, noi, sums, noisy, clean):
            t.record_stream(s)
    return noisy, clean


def batch_norm_snr_iter(records, noise_record, batch_size, snr, L=256, rng=None, device="cuda:0"):
    """Mirror of the reference generator of the same name (local_utils.py:116-130) over in-memory records: `records`
    is an iterable of (650000, leads) arrays, `noise_record` one (>= 650000, leads) array; every full segment of
    `L * batch_size` samples yields (noisy, clean) device tensors of shape (batch_size, leads, L).  The noise offset is
    drawn like the reference's `random.randint(0, 650000 - len - 1)` from `rng` (a `random.Random`)."""
    import random
    import torch
    rng = rng or random.Random()
    noise_dev = torch.as_tensor(noise_record).to(device)
    seg = L * batch_size
    for rec in records:
        rec_dev = torch.as_tensor(rec).to(device)
        n = rec_dev.shape[0]
        for i in range(0, n, seg):
            if i + seg > n:
                break
            j = rng.randint(0, noise_dev.shape[0] - seg - 1)
            yield prep_windows(rec_dev[i:i + seg], noise_dev[j:j + seg], snr, L)
