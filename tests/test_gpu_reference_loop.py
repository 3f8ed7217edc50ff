"""The reference training loop AS WRITTEN (denoise_train.py:24,44-59,66-76: `optim.Adam(model.parameters())`,
`optimizer.zero_grad(); pre = model(data); loss = F.mse_loss(pre, target); loss.backward(); optimizer.step()`, metrics on the
grad-tracked output, an eval pass without no_grad) on the HIP engines with `autograd=True`: the loop below is typed in from
that protocol, nothing of the reference is imported.  It must give what the fused path (`train_step`) and the reference's own
golden loss trajectory give."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
import torch.optim as optim

import ralenet_oracle as O
from parity_util import rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def snr_metric(y1, y2):          # local_utils/evaluate.py:10-29 semantics (dB per window over all leads and samples)
    n1 = (y1 ** 2).flatten(1).sum(1); n2 = ((y1 - y2) ** 2).flatten(1).sum(1)
    return 10 * torch.log10(n1 / n2)


def reference_style_loop(model, batches, epochs=1):
    model = model.cuda()
    optimizer = optim.Adam(model.parameters(), lr=0.001)
    losses, snrs, eval_losses = [], [], []
    for _ in range(epochs):
        model.train()
        for data, target in batches:
            data, target = data.cuda(), target.cuda()
            optimizer.zero_grad()
            pre = model(data)
            loss = F.mse_loss(pre, target)
            losses.append(loss.item())
            loss.backward()
            optimizer.step()
            snrs.append(snr_metric(target, pre))
        model.eval()
        for data, target in batches:
            data, target = data.cuda(), target.cuda()
            pre = model(data)
            eval_losses.append(F.mse_loss(pre, target).item())
    return losses, torch.cat(snrs, dim=0), eval_losses


def test_reference_loop_unchanged_matches_fused_path_and_reference_golden(golden_dir):
    from ecg_denoise_amd import RALENet
    g = np.load(os.path.join(golden_dir, "g3_full_l2_L256.npz"))
    p = O.init_params(O.ralenet_param_shapes("full", 2), 1234)
    x = torch.tensor(g["x"]); tgt = torch.tensor(g["target"])
    m = RALENet("full", leads=2, L=256, max_batch=x.shape[0], device=DEV, autograd=True)
    m.load_state_dict(p, strict=False)
    losses, snrs, eval_losses = reference_style_loop(m, [(x, tgt)] * 3)
    np.testing.assert_allclose(losses, g["adam_losses"], rtol=5e-4)            # the reference's own three Adam steps
    # the fused path from the same start: same trajectory, same weights after three steps
    m2 = RALENet("full", leads=2, L=256, max_batch=x.shape[0], device=DEV)
    m2.load_state_dict(p, strict=False)
    m2.train()
    fused = [m2.train_step(x.to(DEV), tgt.to(DEV))["loss"].item() for _ in range(3)]
    np.testing.assert_allclose(losses, fused, rtol=2e-5)
    sd, sd2 = m.state_dict(), m2.state_dict()
    for k in ("conv1.0.weight", "transformer.blocks.0.mlp.fc1.weight", "utransformer1.blocks.1.attn.proj.bias",
              "rwattn1.relative_position_bias_table", "conv1.2.running_var"):
        assert rel(sd[k].cpu().numpy(), sd2[k].cpu().numpy()) < 2e-4, k
    assert int(sd["conv1.2.num_batches_tracked"]) == 3
    assert snrs.shape == (3 * x.shape[0],) and torch.isfinite(snrs).all()
    assert len(eval_losses) == 3 and all(np.isfinite(eval_losses))
    # .grad of every parameter is a view of the flat gradient buffer, parameters are leaves over the flat parameter buffer
    lo, hi = m.eng.grads.data_ptr(), m.eng.grads.data_ptr() + 4 * m.eng.grads.numel()
    for q in m.parameters():
        assert q.is_leaf and q.requires_grad and lo <= q.grad.data_ptr() < hi


def test_reference_loop_unchanged_on_the_unet_and_an_input_gradient():
    from ecg_denoise_amd import RALENet, UNet
    g_ = torch.Generator().manual_seed(5)
    x = torch.randn(6, 2, 256, generator=g_); tgt = torch.randn(6, 2, 256, generator=g_)
    a, b = UNet(leads=2, L=256, max_batch=6, device=DEV, seed=3, autograd=True), UNet(leads=2, L=256, max_batch=6, device=DEV, seed=3)
    losses, _, _ = reference_style_loop(a, [(x, tgt)] * 3)
    b.train()
    fused = [b.train_step(x.to(DEV), tgt.to(DEV))["loss"].item() for _ in range(3)]
    np.testing.assert_allclose(losses, fused, rtol=2e-5)
    # d loss / d input through the shim = the library's own input gradient
    m = RALENet("full", leads=2, L=256, max_batch=6, device=DEV, seed=7, autograd=True)
    m.train()
    xin = x.to(DEV).requires_grad_(True)
    F.mse_loss(m(xin), tgt.to(DEV)).backward()
    m2 = RALENet("full", leads=2, L=256, max_batch=6, device=DEV, seed=7)
    m2.train()
    y2 = m2(x.to(DEV)); m2.loss_and_metrics(y2, tgt.to(DEV))
    dx = m2.backward(want_dx=True)
    assert rel(xin.grad.cpu().numpy(), dx.cpu().numpy()) < 1e-5
    assert rel(m.eng.grads.cpu().numpy(), m2.eng.grads.cpu().numpy()) < 1e-5
    with torch.no_grad():
        assert m(x.to(DEV)).grad_fn is None


def test_unet_backward_with_a_rescaled_dy_recomputes_its_sums():
    """`m.backward(m._dy.mul_(k))` after forward_loss: a dy passed explicitly is summed again (the sums forward_loss left are
    only used by `backward()` with no argument), so the BatchNorm backward stays consistent: gradients scale by k."""
    from ecg_denoise_amd import UNet
    g_ = torch.Generator().manual_seed(9)
    x = torch.randn(8, 2, 256, generator=g_).to(DEV); tgt = torch.randn(8, 2, 256, generator=g_).to(DEV)
    m = UNet(leads=2, L=256, max_batch=8, device=DEV, seed=4)
    m.train()
    m.forward_loss(x, tgt); m.backward()
    g1 = m.eng.grads.clone()
    m.forward_loss(x, tgt); m.backward(m._dy.mul_(3.0))
    assert rel(m.eng.grads.cpu().numpy(), 3.0 * g1.cpu().numpy()) < 2e-5
