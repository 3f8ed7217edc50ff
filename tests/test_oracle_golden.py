"""The oracle (oracle/ralenet_oracle.py) replayed against the golden vectors that
oracle/gen_golden.py produced from the reference itself.  CPU only."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

import ralenet_oracle as O

NSAMP = 16


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def sample_idx(n):
    return np.unique(np.linspace(0, n - 1, NSAMP).astype(np.int64))


def summarize(named):
    norms, samp = [], []
    for k, t in named.items():
        a = t.detach().double().reshape(-1).numpy()
        norms.append(np.sqrt((a * a).sum()))
        s = a[sample_idx(a.size)] if a.size > NSAMP else a
        samp.append(np.pad(s, (0, NSAMP - s.size)))
    return np.array(norms), np.stack(samp)


def test_pe_tables(golden_dir):
    g = load(golden_dir, "pe_tables")
    for C in O.CHANNELS:
        assert np.array_equal(g[f"C{C}"], O.pe_table(64, C).numpy())


@pytest.mark.parametrize("C,N", [(8, 64), (16, 32), (128, 16)])
@pytest.mark.parametrize("le", [0, 1])
@pytest.mark.parametrize("msk", [0, 1])
def test_block(golden_dir, C, N, le, msk):
    g = load(golden_dir, "g1_blocks")
    tag = f"C{C}_N{N}_le{le}_m{msk}"
    keys = [str(k) for k in g[tag + "_keys"]]
    shapes = OrderedDict()
    probe = O.ralenet_param_shapes("full" if le else "mlp")
    for k in keys:
        base = k[4:]
        lvl = O.CHANNELS.index(C)
        stage = ["dtransformer1", "dtransformer2", "dtransformer3", "dtransformer34", "transformer"][lvl]
        shapes[k] = probe[f"{stage}.blocks.0.{base}"]
    p = O.init_params(shapes, 100 + C + le)
    p = OrderedDict((k, v.requires_grad_(True)) for k, v in p.items())
    x = torch.tensor(g[tag + "_x"], requires_grad=True)
    bias = None
    if msk:
        bias = O.rwave_bias(torch.tensor(g[tag + "_table"]), min(8, N), N)
    y = O.transformer_block(x, p, "blk.", bool(le), bias)
    assert rel(y.detach().numpy(), g[tag + "_y"]) < 2e-6
    (y * torch.tensor(g[tag + "_w"])).sum().backward()
    assert rel(x.grad.numpy(), g[tag + "_dx"]) < 1e-5
    gn = np.array([p[k].grad.double().norm().item() for k in keys])
    np.testing.assert_allclose(gn, g[tag + "_gnorm"], rtol=1e-4)


def test_patch_modules(golden_dir):
    g = load(golden_dir, "g2_modules")
    ppm = O.init_params(OrderedDict([("pm.reduction.weight", (32, 32)), ("pm.norm.weight", (32,)),
                                     ("pm.norm.bias", (32,))]), 7)
    x = torch.tensor(g["pm_x"], requires_grad=True)
    y = O.patch_merge(x, ppm, "pm")
    assert rel(y.detach().numpy(), g["pm_y"]) < 2e-6
    (y * torch.tensor(g["pm_w"])).sum().backward()
    assert rel(x.grad.numpy(), g["pm_dx"]) < 1e-5
    pps = O.init_params(OrderedDict([("ps.reduction.weight", (16, 16)), ("ps.norm.weight", (16,)),
                                     ("ps.norm.bias", (16,))]), 7)
    x = torch.tensor(g["ps_x"], requires_grad=True)
    y = O.patch_separate(x, pps, "ps")
    assert rel(y.detach().numpy(), g["ps_y"]) < 2e-6
    (y * torch.tensor(g["ps_w"])).sum().backward()
    assert rel(x.grad.numpy(), g["ps_dx"]) < 1e-5
    b = O.rwave_bias(torch.tensor(g["rw_table"]), 8, 32)
    assert np.array_equal(b.numpy(), g["rw_bias"])


# (the last three: window lengths that are multiples of 16 but not of 256 - oracle/gen_golden_r6.py)
CASES = [("nra", 2, 512), ("nra", 2, 256), ("full", 2, 256), ("mlp", 2, 256), ("full", 2, 512),
         ("full", 1, 512), ("full", 2, 1024), ("nra", 2, 320), ("nra", 2, 128), ("full", 2, 640)]


@pytest.mark.parametrize("variant,leads,L", CASES)
def test_ralenet_whole_model(golden_dir, variant, leads, L):
    g = load(golden_dir, f"g3_{variant}_l{leads}_L{L}")
    p = O.init_params(O.ralenet_param_shapes(variant, leads), 1234)
    assert [str(k) for k in g["keys"]] == list(p.keys())
    x = torch.tensor(g["x"]); tgt = torch.tensor(g["target"])
    bn = O.new_bn_state()
    m = OrderedDict((k, torch.zeros_like(v)) for k, v in p.items())
    v = OrderedDict((k, torch.zeros_like(t)) for k, t in p.items())
    p0 = OrderedDict((k, t.clone()) for k, t in p.items())
    fwd = lambda pp, xx: O.ralenet_forward(pp, xx, variant, True, bn)
    r = O.train_step(p, x, tgt, fwd, m, v, 1)
    assert rel(r["pred"].numpy(), g["y_train"]) < 5e-6
    assert abs(r["loss"].item() - g["loss"]) < 1e-5 * abs(g["loss"])
    gn, gs = summarize(r["grads"])
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=2e-4, atol=1e-9)
    assert rel(gs, g["grad_samp"]) < 2e-4
    np.testing.assert_allclose(bn["running_mean"].numpy(), g["bn_mean_conv1.2"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(bn["running_var"].numpy(), g["bn_var_conv1.2"], rtol=1e-5, atol=1e-7)
    # to_kv.bias[:C] (the key bias) has an exactly-zero true gradient (softmax is
    # shift invariant), so its computed gradient is rounding noise that Adam turns into
    # +-lr steps: compare parameter norms at lr granularity only.
    a1n, a1s = summarize(p)
    np.testing.assert_allclose(a1n, g["adam1_norm"], rtol=1e-4)
    assert rel(a1s, g["adam1_samp"]) < 1e-4
    # eval forward with the post-step-1 running stats but the ORIGINAL weights
    with torch.no_grad():
        ye = O.ralenet_forward(p0, x, variant, False, bn)
    assert rel(ye.numpy(), g["y_eval"]) < 5e-6
    losses = [r["loss"].item()]
    for s in (2, 3):
        losses.append(O.train_step(p, x, tgt, fwd, m, v, s)["loss"].item())
    np.testing.assert_allclose(losses, g["adam_losses"], rtol=2e-4)
    a3n, a3s = summarize(p)
    np.testing.assert_allclose(a3n, g["adam3_norm"], rtol=5e-4)


def test_unet_whole_model(golden_dir):
    g = load(golden_dir, "g3_unet_l2_L512")
    p = O.init_params(O.unet_param_shapes(), 1234)
    x = torch.tensor(g["x"]); tgt = torch.tensor(g["target"])
    bn = O.unet_bn_state(p)
    m = OrderedDict((k, torch.zeros_like(t)) for k, t in p.items())
    v = OrderedDict((k, torch.zeros_like(t)) for k, t in p.items())
    p0 = OrderedDict((k, t.clone()) for k, t in p.items())
    fwd = lambda pp, xx: O.unet_forward(pp, xx, True, bn)
    r = O.train_step(p, x, tgt, fwd, m, v, 1)
    assert rel(r["pred"].numpy(), g["y_train"]) < 5e-6
    gn, gs = summarize(r["grads"])
    # conv biases in front of a BatchNorm have an exactly-zero true gradient: their computed value is
    # rounding noise (~1e-8) that depends on the thread count, hence the absolute floor
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=2e-4, atol=1e-6)
    for k in O.UNET_BN:
        np.testing.assert_allclose(bn[k]["running_var"].numpy(), g["bn_var_" + k], rtol=1e-5, atol=1e-7)
    with torch.no_grad():
        ye = O.unet_forward(p0, x, False, bn)
    assert rel(ye.numpy(), g["y_eval"]) < 5e-6
    a1n, _ = summarize(p)
    keep = np.array([not (k.endswith("conv.bias")) for k in p])   # zero-gradient biases take +-lr noise steps
    np.testing.assert_allclose(a1n[keep], g["adam1_norm"][keep], rtol=1e-5)


def test_acdae(golden_dir):
    """the ACDAE restatement against the reference's own outputs, gradients and Adam trajectory"""
    g = load(golden_dir, "g3_acdae_l2_L512")
    p = O.init_params(O.acdae_param_shapes(), 1234)
    assert [str(k) for k in g["keys"]] == list(p.keys())
    x = torch.tensor(g["x"]); tgt = torch.tensor(g["target"])
    m = OrderedDict((k, torch.zeros_like(v)) for k, v in p.items())
    v = OrderedDict((k, torch.zeros_like(t)) for k, t in p.items())
    fwd = lambda pp, xx: O.acdae_forward(pp, xx)
    losses = []
    for s_ in (1, 2, 3):
        r = O.train_step(p, x, tgt, fwd, m, v, s_)
        losses.append(r["loss"].item())
        if s_ == 1:
            assert rel(r["pred"].numpy(), g["y_train"]) < 5e-6
            assert rel(r["pred"].numpy(), g["y_eval"]) < 5e-6          # no BatchNorm: train == eval
            gn, gs = summarize(r["grads"])
            np.testing.assert_allclose(gn, g["grad_norm"], rtol=2e-4, atol=1e-9)
            assert rel(gs, g["grad_samp"]) < 2e-4
            a1n, _ = summarize(p)
            np.testing.assert_allclose(a1n, g["adam1_norm"], rtol=1e-4)
    np.testing.assert_allclose(losses, g["adam_losses"], rtol=2e-4)


def test_newrale(golden_dir):
    g = load(golden_dir, "g3_newrale_L256")
    p = O.init_params(O.ralenet_param_shapes("full", 2), 1234)
    pa = O.init_params(O.newrale_param_shapes(), 77)
    pa = OrderedDict((k, t.requires_grad_(True)) for k, t in pa.items())
    x = torch.tensor(g["x"]); tgt = torch.tensor(g["target"])
    y = O.newrale_forward(pa, p, x, "full", True, O.new_bn_state())
    assert rel(y.detach().numpy(), g["y_train"]) < 5e-6
    O.mse(y, tgt).backward()
    for k in pa:
        assert rel(pa[k].grad.numpy(), g["grad_" + k]) < 2e-4, k


def test_metrics(golden_dir):
    g = load(golden_dir, "g4_metrics")
    y = torch.tensor(g["y"]); pred = torch.tensor(g["pred"])
    np.testing.assert_allclose(O.snr(y, pred).numpy(), g["snr"], rtol=1e-6)
    np.testing.assert_allclose(O.rmse(y, pred).numpy(), g["rmse"], rtol=1e-6)
    np.testing.assert_allclose(O.snr(y, 0.9 * y).numpy(), 20.0, atol=1e-4)
    np.testing.assert_allclose(g["snr_09"], 20.0, atol=1e-4)


def test_reference_run_to_run_spread_fixtures(golden_dir):
    """What the 100-epoch curves of the REFERENCE (thread counts 6 / 2 / 3 / 4 / 1) say about its own repeatability (same data, same initial weights,
    only the intra-op thread count = summation order differs; oracle/gen_ref_train_curve.py): identical to 1e-3 dB for
    two epochs, apart by more than 0.1 dB from epoch 3, 0.55 dB apart at the end.  The north star's "within 0.05 dB of
    the reference" is therefore testable for the first ~500 optimiser steps only; after that the bar is the reference's
    own spread (tests/test_gpu_train_loop.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(golden_dir, "g6_ref_train_curve_full*.npz")))
    c = np.stack([np.load(f)["test_snr"] for f in files])
    assert c.shape[0] >= 4 and c.shape[1] == 100
    assert np.ptp(c[:, :2], axis=0).max() < 1e-3
    assert np.ptp(c[:, 2]) > 0.1
    assert np.ptp(c[:, -1]) > 0.4
    assert 0.2 < np.ptp(c[:, -10:].mean(1)) < 0.6


def test_danet_oracle_matches_reference_golden(golden_dir):
    """oracle/danet_oracle.py against the reference's model/DAM.py::Seq2Seq2 (fixture made by oracle/gen_golden_danet.py):
    eval output, train output, loss, every parameter gradient, every running statistic after one training forward -
    including the DAM's shared fcn, whose BatchNorm statistics are updated twice per forward."""
    import danet_oracle as D
    g = np.load(os.path.join(golden_dir, "g3_danet_L512.npz"))
    x = torch.from_numpy(g["x"]).double(); tgt = torch.from_numpy(g["target"]).double()
    st = D.init_state(4321, dtype=torch.float64)
    with torch.no_grad():
        ye = D.danet_forward(st, x, training=False)
    assert rel(ye.numpy(), g["y_eval"]) < 1e-5
    params = OrderedDict((k, v.requires_grad_(True)) for k, v in st.items() if D.is_param(k) and ".dam.fcn2." not in k)
    y = D.danet_forward(st, x, training=True)
    loss = torch.nn.functional.mse_loss(y, tgt)
    loss.backward()
    assert rel(y.detach().numpy(), g["y_train"]) < 1e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    for k, p in params.items():
        if k.endswith((".fcn.0.bias", ".fcn.3.bias", ".fcn1.0.bias", ".fcn1.3.bias")):
            # a bias in front of a batch-statistics BatchNorm has no gradient: the reference holds fp32 rounding noise
            assert np.abs(g["grad_" + k]).max() < 1e-5 and np.abs(p.grad.numpy()).max() < 1e-12, k
        else:
            assert rel(p.grad.numpy(), g["grad_" + k]) < 2e-4, k
    n = 0
    for k in st:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel(st[k].numpy(), g["after_" + k]) < 1e-5, k
            n += 1
        if k.endswith("num_batches_tracked"):
            assert int(st[k]) == int(g["after_" + k]) == (2 if ".dam.fcn" in k else 1), k
    assert n == 2 * (4 * 3 + 4 * 3 + 3 * 4)


# ---------------------------------------------------------------------------------------------------------------------
# wavelet-threshold baseline: the oracle against the reference function run on PyWavelets 1.1.1 (oracle/gen_golden_wavelet.py)
# ---------------------------------------------------------------------------------------------------------------------
def test_wavelet_oracle_matches_the_reference_function_on_pywavelets(golden_dir):
    import baselines_oracle as W
    g = np.load(os.path.join(golden_dir, "g8_wavelet.npz"))
    assert str(g["pywt_version"]) == "1.1.1"
    # the pieces, one by one
    for k, mine in (("dec_lo", W.DB8_DEC_LO), ("dec_hi", W.DB8_DEC_HI), ("rec_lo", W.DB8_REC_LO), ("rec_hi", W.DB8_REC_HI)):
        assert np.array_equal(g[k], mine), k
    assert [W.dwt_max_level(int(n)) for n in g["maxlev_n"]] == [int(v) for v in g["maxlev"]]
    co = W.wavedec(g["x_512"][0], 5)
    assert [len(c) for c in co] == [int(v) for v in g["bands_len"]]                 # 30 30 46 77 139 263
    assert np.abs(np.concatenate(co) - g["bands_flat"]).max() < 1e-12              # boundary mode = half-sample symmetric
    assert np.abs(W.waverec(co) - g["waverec"]).max() < 1e-12
    co = W.wavedec(g["x_300"][1], 4)                                               # odd band lengths on the way down
    assert [len(c) for c in co] == [int(v) for v in g["bands300_len"]]
    assert np.abs(np.concatenate(co) - g["bands300_flat"]).max() < 1e-12
    assert np.abs(W.waverec(co) - g["waverec300"]).max() < 1e-12
    assert np.array_equal(W.threshold_soft(g["thr_in"], 0.5), g["thr_out"])
    # the whole function (denoisefunc.py:7-33): 2-D inputs of four lengths, a 3-D input, a float32 input
    for n in (256, 300, 512, 1024):
        assert np.abs(W.wavelet_denoise(g[f"x_{n}"]) - g[f"y_{n}"]).max() < 1e-12, n
    assert np.abs(W.wavelet_denoise(g["x3_512"]) - g["y3_512"]).max() < 1e-12
    assert g["y32_512"].dtype == np.float32                                        # (pywt computes float32 inputs in float32)
    assert np.abs(W.wavelet_denoise(g["x32_512"]) - g["y32_512"]).max() < 5e-6
    # documented divergence: a record whose detail bands are exactly zero is NaN in pywt (0 / 0 in its soft threshold)
    assert np.isnan(g["y_zero"]).all() and np.array_equal(W.wavelet_denoise(np.zeros((1, 512))), np.zeros((1, 512)))
