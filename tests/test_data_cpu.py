"""Data path (reference on-disk layout) and synthetic generator — CPU only."""
import os

import numpy as np

from ecg_denoise_amd import data, synth


def test_synthetic_dataset_layout_and_snr(tmp_path):
    p = data.write_synthetic_dataset(str(tmp_path / "dict_data"), n=64, L=256, noise_types=["bw", "emb"], intensities=[-4, 0])
    assert sorted(os.listdir(p)) == ["0", "ecg.npy", "m4"]
    ds = data.EcgDataset("emb", 0, p)
    assert len(ds) == 64 and ds[3][0].shape == (2, 256) and ds[3][1].shape == (2, 256)
    noisy, clean = ds.data, ds.ground_data
    snr = 10 * np.log10((clean ** 2).mean((1, 2)) / ((noisy - clean) ** 2).mean((1, 2)))
    np.testing.assert_allclose(snr, 0.0, atol=1e-3)                 # single_snr_noise_add scaling
    np.testing.assert_allclose(clean.std(-1), 1.0, atol=1e-4)        # np_norm
    ds4 = data.EcgDataset(["bw", "emb"], -4, p)
    assert len(ds4) == 128
    snr4 = 10 * np.log10((ds.ground_data ** 2).mean((1, 2)) / ((ds4.data[:64] - ds.ground_data) ** 2).mean((1, 2)))
    np.testing.assert_allclose(snr4, -4.0, atol=1e-3)
    tr, te = data.make_loaders(ds, batch_size=16, n_select=40)
    assert len(tr.dataset) == 32 and len(te.dataset) == 8 and len(tr) == 2
    xb, yb = next(iter(tr))
    assert xb.dtype == np.float32 and xb.shape == (16, 2, 256) and yb.shape == (16, 2, 256)
    # pairs stay aligned through selection / shuffling
    d = ((xb - yb) ** 2).mean((1, 2))
    np.testing.assert_allclose(10 * np.log10((yb ** 2).mean((1, 2)) / d), 0.0, atol=1e-3)


def test_generator_is_deterministic_and_r_peak_is_central():
    a, ca = synth.make_dataset(8, 2, 512, "ma", 2.0, seed=5)
    b, cb = synth.make_dataset(8, 2, 512, "ma", 2.0, seed=5)
    assert np.array_equal(a, b) and np.array_equal(ca, cb)
    centre = np.abs(ca[:, 0, 256 - 40:256 + 40]).max(-1)         # one R peak sits in the central eighth
    assert np.all(centre >= 0.9 * np.abs(ca[:, 0]).max(-1))


def test_scoring_functions_match_reference_arithmetic():
    """test_cls.py:14-29 on a hand-checkable case: labels 1 1 1 0 0 0, predictions 1 1 0 1 0 0 -> TP 2, FP 1, FN 1"""
    import torch
    from ecg_denoise_amd import scoring
    label = torch.tensor([1, 1, 1, 0, 0, 0])
    logits = torch.tensor([[0., 1.], [0., 2.], [1., 0.], [0., 1.], [3., 0.], [1., 0.]])
    assert scoring.acc(logits, label) == 4 / 6
    assert scoring.precision(logits, label) == 2 / 3
    assert scoring.f1_score(logits, label) == 2 / (2 + 0.5 * 2)
    loader = [(torch.zeros(3, 2, 8), label[:3]), (torch.zeros(3, 2, 8), label[3:])]
    it = iter([logits[:3], logits[3:]])
    a, p, f = scoring.score_denoiser(lambda d: next(it), None, loader, "cpu")
    assert (a, p, f) == (4 / 6, 2 / 3, 2 / 3)
