"""The host training loop (ecg_denoise_amd/train.py) replayed against the trace that the
reference's own denoise_train.train produced on the same arrays (tests/golden/g5_train_trace.npz)."""
import os

import numpy as np
import pytest
import torch

import ralenet_oracle as O
from parity_util import rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def batches(x, y, bs):
    return [(x[i:i + bs], y[i:i + bs]) for i in range(0, len(x), bs)]


def test_train_loop_reproduces_reference_trace(golden_dir, tmp_path):
    from ecg_denoise_amd import RALENet
    from ecg_denoise_amd.train import train
    g = np.load(os.path.join(golden_dir, "g5_train_trace.npz"))
    clean, noisy = torch.tensor(g["clean"]), torch.tensor(g["noisy"])
    m = RALENet("nra", leads=2, L=256, max_batch=32, device="cuda:0")
    m.load_state_dict(O.init_params(O.ralenet_param_shapes("nra", 2), 4321), strict=False)
    res = train(epochs=10, model=m, batch_size=32, train_loader=batches(noisy[:96], clean[:96], 32),
                test_loader=batches(noisy[96:], clean[96:], 32), use_gpu=True, model_name="ralenet_nra",
                noise_name="emb", noise_intensity=0, out_dir=str(tmp_path), log=lambda *_: None)
    # 30 Adam steps of a chaotic system in fp32: trajectories agree to ~1e-3 in dB
    np.testing.assert_allclose(res[0], g["train_snr"], atol=5e-3, rtol=2e-3)
    np.testing.assert_allclose(res[1], g["test_snr"], atol=5e-3, rtol=2e-3)
    np.testing.assert_allclose(res[2], g["train_rmse"], rtol=2e-3)
    np.testing.assert_allclose(res[3], g["test_rmse"], rtol=2e-3)
    saved = sorted(os.listdir(tmp_path / "model_save" / "ralenet_nra"))
    assert saved == [str(s) for s in g["saved"]]
    line = open(tmp_path / "output.txt").read()
    ref = str(g["output_line"])
    assert line.split(":snr:")[0] == ref.split(":snr:")[0]
    assert abs(float(line.split(":snr:")[1].split(",")[0]) - float(ref.split(":snr:")[1].split(",")[0])) < 5e-3
    # the checkpoint is a plain reference-format state_dict and resumes
    sd = torch.load(tmp_path / "model_save" / "ralenet_nra" / saved[0])
    assert set(sd) == set(m.state_dict()) and int(sd["conv1.2.num_batches_tracked"]) == 30
    m2 = RALENet("nra", leads=2, L=256, max_batch=32, device="cuda:0", seed=0)
    r2 = train(epochs=1, model=m2, batch_size=32, train_loader=[], test_loader=batches(noisy[96:], clean[96:], 32),
               model_path=str(tmp_path / "model_save" / "ralenet_nra" / saved[0]), model_name="resumed",
               noise_name="emb", noise_intensity=0, out_dir=str(tmp_path / "r"), log=lambda *_: None)
    assert abs(r2[1][0] - res[1][-1]) < 1e-5      # resumed weights + running stats give the same test SNR


def test_full_protocol_first_epochs_match_reference_curve(golden_dir, tmp_path):
    """main.py protocol (10 000 synthetic windows, 8000/2000 split, batch 32, Adam 1e-3, emb noise at 0 dB) on the
    'full' RA-LENet: the first two epochs (500 optimiser steps) must reproduce the per-epoch SNR the REFERENCE itself
    reached on the same arrays and initial weights (g6_ref_train_curve_full.npz, oracle/gen_ref_train_curve.py) to well
    inside the 0.05 dB the north star allows; later epochs diverge chaotically (tools/snr_experiment.py)."""
    from ecg_denoise_amd import RALENet, synth
    from ecg_denoise_amd.train import train
    g = np.load(os.path.join(golden_dir, "g6_ref_train_curve_full.npz"))
    noisy, clean = synth.make_dataset(10000, 2, 256, "emb", 0.0, seed=2023)
    (trn, trc), (ten, tec) = synth.split_8000_2000(noisy, clean)
    m = RALENet("full", leads=2, L=256, max_batch=32, device="cuda:0")
    sd = O.init_params(O.ralenet_param_shapes("full", 2), int(g["seed"]))
    for k in sd:   # reference default scale of the tensors the build's init rule perturbs
        if "relative_position_bias_table" in k:
            sd[k].zero_()
        elif ".norm" in k or k.startswith("conv1.2."):
            sd[k].fill_(1.0 if k.endswith("weight") else 0.0)
    m.load_state_dict(sd, strict=False)
    res = train(epochs=2, model=m, batch_size=32, train_loader=batches(trn, trc, 32), test_loader=batches(ten, tec, 32),
                use_gpu=True, model_name="ralenet", noise_name="emb", noise_intensity=0, out_dir=str(tmp_path),
                log=lambda *_: None)
    assert np.abs(np.array(res[1]) - g["test_snr"][:2]).max() < 0.02, (res[1], g["test_snr"][:2])
    assert np.abs(np.array(res[0]) - g["train_snr"][:2]).max() < 0.02, (res[0], g["train_snr"][:2])


def test_full_protocol_end_of_training_inside_the_reference_spread(golden_dir, tmp_path):
    """The whole 100-epoch main.py protocol (64 s on one MI355X) against the reference's own run-to-run spread: the
    reference fixtures (same data, same initial weights, intra-op thread counts 6 / 2 / 3 / 4 / 1, i.e. different summation
    orders: g6_ref_train_curve_full*.npz) end with last-ten-epoch means 19.36 .. 19.69 dB (mean 19.56, s.d. 0.14) after
    agreeing to 1e-3 dB for two epochs.  Twelve recorded HIP runs from the same weights: mean 19.58, s.d. 0.14
    (profiles/r02_snr_experiment.json).  The bar: the mean of TWO HIP runs (s.d. 0.10) inside the reference band widened
    by one reference standard deviation, and within 0.3 dB (two and a half combined s.d.) of the reference mean."""
    from ecg_denoise_amd import RALENet, synth
    from ecg_denoise_amd.train import train
    import glob
    refs = [np.load(f) for f in sorted(glob.glob(os.path.join(golden_dir, "g6_ref_train_curve_full*.npz")))]
    assert len(refs) >= 4
    last10 = np.array([r["test_snr"][-10:].mean() for r in refs])
    noisy, clean = synth.make_dataset(10000, 2, 256, "emb", 0.0, seed=2023)
    (trn, trc), (ten, tec) = synth.split_8000_2000(noisy, clean)
    sd = O.init_params(O.ralenet_param_shapes("full", 2), int(refs[0]["seed"]))
    for k in sd:
        if "relative_position_bias_table" in k:
            sd[k].zero_()
        elif ".norm" in k or k.startswith("conv1.2."):
            sd[k].fill_(1.0 if k.endswith("weight") else 0.0)
    runs = []
    for r in range(2):
        m = RALENet("full", leads=2, L=256, max_batch=32, device="cuda:0")
        m.load_state_dict(sd, strict=False)
        res = train(epochs=100, model=m, batch_size=32, train_loader=batches(trn, trc, 32), test_loader=batches(ten, tec, 32),
                    use_gpu=True, model_name="ralenet", noise_name="emb", noise_intensity=0, out_dir=str(tmp_path / str(r)),
                    log=lambda *_: None)
        runs.append(float(np.mean(res[1][-10:])))
        del m
    mine = float(np.mean(runs))
    s = float(last10.std(ddof=1))
    assert last10.min() - s <= mine <= last10.max() + s, (runs, last10)
    assert abs(mine - last10.mean()) < 0.3, (runs, last10)

