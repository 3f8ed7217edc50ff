"""ral_prep_windows (GPU windowing + noise mixing, through the C ABI) against the oracle and the reference vectors."""
import os
import random

import numpy as np
import pytest
import torch

import dataprep_oracle as D

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_matches_reference_vectors(golden_dir):
    from ecg_denoise_amd.data import prep_windows
    g = np.load(os.path.join(golden_dir, "g7_dataprep.npz"))
    for n in "abcd":
        noisy, clean = prep_windows(torch.tensor(g[n + "_sig"], device=DEV), torch.tensor(g[n + "_noise"], device=DEV),
                                    float(g[n + "_snr"]), int(g[n + "_L"]))
        # double statistics + one rounding, like the reference: at most the last fp32 bit differs (summation order)
        np.testing.assert_allclose(noisy.cpu().numpy(), g[n + "_noisy"], rtol=3e-7, atol=3e-7)
        np.testing.assert_allclose(clean.cpu().numpy(), g[n + "_clean"], rtol=3e-7, atol=3e-7)


def test_full_record_segment_matches_oracle_and_properties():
    """BASELINE-size segment (a 650 000-sample record cut to 2539 windows of 256): oracle equality plus the
    size-independent properties (unit variance per lead, requested SNR, window order)."""
    from ecg_denoise_amd.data import prep_windows
    rng = np.random.default_rng(7)
    T, L = 2539 * 256, 256
    sig = (1024 + 300 * np.sin(np.arange(T)[:, None] / 57.0) + 60 * rng.standard_normal((T, 2))).astype(np.int32)
    noise = (25 * rng.standard_normal((T, 2))).astype(np.int32)
    noisy, clean = prep_windows(torch.tensor(sig, device=DEV), torch.tensor(noise, device=DEV), -2.0, L)
    on, oc = D.prep_segment(sig, noise, -2.0, L)
    np.testing.assert_allclose(noisy.cpu().numpy(), on, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(clean.cpu().numpy(), oc, rtol=1e-6, atol=1e-6)
    c = clean.double()
    assert abs(10 * torch.log10((c ** 2).mean() / ((noisy.double() - c) ** 2).mean()).item() + 2.0) < 1e-4
    flat = c.permute(1, 0, 2).reshape(2, -1)
    assert flat.mean(1).abs().max().item() < 1e-6 and (flat.std(1, unbiased=False) - 1).abs().max().item() < 1e-6
    # window b, sample l of lead c is segment row b * L + l
    assert torch.equal(clean[5, 1, :3].cpu(), torch.tensor(oc[5, 1, :3]))


def test_iterator_mirrors_reference_generator():
    from ecg_denoise_amd.data import batch_norm_snr_iter
    rng = np.random.default_rng(3)
    rec = (1000 + 100 * rng.standard_normal((256 * 32 * 3 + 100, 2))).astype(np.int32)
    noise = (20 * rng.standard_normal((650000, 2))).astype(np.int32)
    out = list(batch_norm_snr_iter([rec], noise, 32, 0.0, L=256, rng=random.Random(500), device=DEV))
    assert len(out) == 3 and out[0][0].shape == (32, 2, 256)      # the ragged tail segment is dropped (`break`)
    r = random.Random(500)
    j = r.randint(0, 650000 - 256 * 32 - 1)
    on, oc = D.prep_segment(rec[:256 * 32], noise[j:j + 256 * 32], 0.0, 256)
    np.testing.assert_allclose(out[0][0].cpu().numpy(), on, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out[0][1].cpu().numpy(), oc, rtol=1e-6, atol=1e-6)


def test_rejects_bad_arguments():
    from ecg_denoise_amd import _lib
    from ecg_denoise_amd.data import prep_windows
    with pytest.raises(_lib.RalError):
        prep_windows(torch.zeros(1000, 2, device=DEV), torch.zeros(1000, 2, device=DEV), 0.0, 256)   # T % L != 0
    with pytest.raises(_lib.RalError):
        prep_windows(torch.zeros(512, 2), torch.zeros(512, 2), 0.0, 256)                           # host tensors
