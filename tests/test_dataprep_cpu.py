"""The windowing / noise-mixing oracle against the vectors the reference's own np_norm / Gnoisegen / rearrange produced
(tests/golden/g7_dataprep.npz, oracle/gen_golden_dataprep.py)."""
import os

import numpy as np

import dataprep_oracle as D


def test_oracle_matches_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_dataprep.npz"))
    for n in "abcd":
        noisy, clean = D.prep_segment(g[n + "_sig"], g[n + "_noise"], float(g[n + "_snr"]), int(g[n + "_L"]))
        assert noisy.dtype == np.float32 and noisy.shape == g[n + "_noisy"].shape
        np.testing.assert_array_equal(noisy, g[n + "_noisy"])
        np.testing.assert_array_equal(clean, g[n + "_clean"])


def test_snr_of_the_mix_is_the_requested_one():
    rng = np.random.default_rng(1)
    sig = rng.standard_normal((1024, 2)) * 50 + 1000
    noise = rng.standard_normal((1024, 2)) * 7
    for snr in (-4.0, 0.0, 2.0):
        noisy, clean = D.prep_segment(sig, noise, snr, 256)
        p_sig = np.mean(clean.astype(np.float64) ** 2)
        p_noise = np.mean((noisy.astype(np.float64) - clean) ** 2)
        assert abs(10 * np.log10(p_sig / p_noise) - snr) < 1e-4
        # per-lead z-score over the whole segment
        flat = clean.transpose(1, 0, 2).reshape(2, -1)
        np.testing.assert_allclose(flat.mean(1), 0.0, atol=1e-5)
        np.testing.assert_allclose(flat.std(1), 1.0, atol=1e-5)
