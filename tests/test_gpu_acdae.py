"""ACDAE comparison baseline (reference model/ACDAE.py) on the HIP path vs the fp64 oracle and the reference's golden
vectors: outputs, loss, every parameter gradient, the input gradient, three Adam steps."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

import ralenet_oracle as O
from parity_util import rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("L,B", [(512, 4), (256, 3), (1024, 2)])
def test_acdae_train_step_matches_oracle(L, B):
    from ecg_denoise_amd import ACDAE
    p32 = O.init_params(O.acdae_param_shapes(), 1234)
    g = torch.Generator().manual_seed(2023)
    x = torch.randn(B, 2, L, generator=g); tgt = torch.randn(B, 2, L, generator=g)
    m = ACDAE(L=L, max_batch=B, device=DEV)
    m.load_state_dict(p32)
    m.train()
    y = m(x.to(DEV))
    loss, snr, rmse = m.loss_and_metrics(y, tgt.to(DEV))
    dx = m.backward(want_dx=True)
    torch.cuda.synchronize()
    p = OrderedDict((k, v.double().requires_grad_(True)) for k, v in p32.items())
    xd = x.double().requires_grad_(True)
    yo = O.acdae_forward(p, xd)
    lo = O.mse(yo, tgt.double())
    grads = torch.autograd.grad(lo, list(p.values()) + [xd])
    assert rel(y.cpu().numpy(), yo.detach().numpy()) < 1e-5
    assert abs(loss.item() - lo.item()) < 1e-5 * abs(lo.item())
    np.testing.assert_allclose(snr.cpu().numpy(), O.snr(tgt.double(), yo.detach()).numpy(), rtol=0, atol=1e-4)
    ng = m.named_grads()
    bad = {k: e for (k, _), gr in zip(p.items(), grads[:-1]) if (e := rel(ng[k].cpu().numpy(), gr.numpy())) > 1e-4}
    assert not bad, bad
    assert rel(dx.cpu().numpy(), grads[-1].numpy()) < 1e-4
    m.eval()
    assert torch.equal(m(x.to(DEV)), y)                    # no BatchNorm, no dropout: eval is the same function


def test_acdae_against_reference_golden(golden_dir):
    from ecg_denoise_amd import ACDAE
    g = np.load(os.path.join(golden_dir, "g3_acdae_l2_L512.npz"))
    p = O.init_params(O.acdae_param_shapes(), 1234)
    x = torch.tensor(g["x"]).to(DEV); tgt = torch.tensor(g["target"]).to(DEV)
    m = ACDAE(L=512, max_batch=x.shape[0], device=DEV)
    assert [k for k, _ in m.named_parameters()] == [str(k) for k in g["keys"]]      # the reference's state_dict order
    m.load_state_dict(p)
    m.train()
    y = m(x)
    loss, _, _ = m.loss_and_metrics(y, tgt)
    m.backward()
    assert rel(y.cpu().numpy(), g["y_train"]) < 1e-5
    assert abs(loss.item() - g["loss"]) < 1e-5 * abs(g["loss"])
    ng = m.named_grads()
    gn = np.array([ng[str(k)].double().norm().item() for k in g["keys"]])
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=2e-4, atol=1e-8)
    m2 = ACDAE(L=512, max_batch=x.shape[0], device=DEV)
    m2.load_state_dict(p)
    losses = [m2.train_step(x, tgt)["loss"].item() for _ in range(3)]
    np.testing.assert_allclose(losses, g["adam_losses"], rtol=5e-4)
    sd = m2.state_dict()
    assert list(sd) == [str(k) for k in g["keys"]] and sd["DecList.0.ECA.conv.weight"].shape == (1, 1, 3)


def test_acdae_trains_through_the_harness(tmp_path):
    """main.py:66-68 path: ACDAE through the same train() harness as the other models; the loss goes down"""
    from ecg_denoise_amd import ACDAE, synth
    from ecg_denoise_amd.train import train
    noisy, clean = synth.make_dataset(96, 2, 256, "emb", 0.0, seed=5)
    bat = lambda a, b, bs: [(a[i:i + bs], b[i:i + bs]) for i in range(0, len(a), bs)]
    m = ACDAE(L=256, max_batch=32, device=DEV, seed=3)
    res = train(epochs=10, model=m, batch_size=32, train_loader=bat(noisy[:64], clean[:64], 32), test_loader=bat(noisy[64:], clean[64:], 32),
                use_gpu=True, model_name="ACDAE", noise_name="emb", noise_intensity=0, out_dir=str(tmp_path), log=lambda *_: None)
    tl, _ = train.last_losses
    assert tl[-1] < tl[0] and all(np.isfinite(res[1]))
    assert (tmp_path / "model_save" / "ACDAE" / "ACDAE_9_emb_intensity0.pth").exists()


def test_acdae_bench_batch_is_the_sum_of_its_halves():
    """BASELINE batch (2048 x 2 x 512).  ACDAE has no BatchNorm, so windows are independent: the forward of a slice is the
    slice of the forward bit for bit, the input gradient likewise, and the parameter gradients of the whole batch are the
    sum of the gradients of its two halves (same upstream gradient) - which exercises the multi-window loops and the
    split-K accumulation of the MFMA weight-gradient kernel at full size.  64 windows of it also against the fp64 oracle."""
    from ecg_denoise_amd import ACDAE
    B, L = 2048, 512
    p32 = O.init_params(O.acdae_param_shapes(), 77)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 2, L, generator=g).to(DEV)
    dy = (torch.randn(B, 2, L, generator=g) / (B * 2 * L)).to(DEV)
    m = ACDAE(L=L, max_batch=B, device=DEV); m.load_state_dict(p32); m.train()
    y = m(x).clone()
    dx = m.backward(dy, want_dx=True).clone()
    gw = OrderedDict((k, v.clone()) for k, v in m.named_grads().items())
    h = ACDAE(L=L, max_batch=B // 2, device=DEV); h.load_state_dict(p32); h.train()
    parts = []
    for lo in (0, B // 2):
        yh = h(x[lo:lo + B // 2].contiguous())
        assert torch.equal(yh, y[lo:lo + B // 2])
        dxh = h.backward(dy[lo:lo + B // 2].contiguous(), want_dx=True)
        assert torch.equal(dxh, dx[lo:lo + B // 2])
        parts.append(OrderedDict((k, v.clone()) for k, v in h.named_grads().items()))
    for k in gw:
        s2 = parts[0][k].double() + parts[1][k].double()
        assert rel(gw[k].double().cpu().numpy(), s2.cpu().numpy()) < 2e-5, k      # (fp32 atomics: the order of a million-term sum)
    # a slice against the oracle
    n = 64
    p = OrderedDict((k, v.double()) for k, v in p32.items())
    yo = O.acdae_forward(p, x[:n].cpu().double())
    assert rel(y[:n].cpu().numpy(), yo.numpy()) < 1e-5
