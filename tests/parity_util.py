"""Shared helpers of the GPU parity tests: run the HIP path and the CPU oracle on the
same seeded inputs and report relative L2 errors per tensor."""
from collections import OrderedDict

import numpy as np
import torch

import ralenet_oracle as O


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).ravel(); b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def oracle_trace(p, x, variant, bn, dtype=torch.float64):
    """Oracle forward (training mode) that also returns every block / resample output,
    named like the library's debug tensors."""
    import math
    import torch.nn.functional as F
    le, rw, _ = O.variant_flags(variant)
    tr = OrderedDict()
    B, _, L = x.shape
    y = F.conv1d(x, p["conv1.0.weight"], p["conv1.0.bias"], padding=1)
    y = F.leaky_relu(y, 0.2)
    tr["a0"] = y.permute(0, 2, 1)
    y = F.batch_norm(y, bn["running_mean"], bn["running_var"], p["conv1.2.weight"], p["conv1.2.bias"], True, 0.1, 1e-5)
    x0 = y
    tr["x0"] = y.permute(0, 2, 1)
    biases = [None] * 5
    if rw:
        for i, ln in enumerate(O.RW_LEN):
            biases[i + 1] = O.rwave_bias(p[f"rwattn{i+1}.relative_position_bias_table"], ln, L >> i)
    bi = [0]

    def stage(t, name, rwi):
        for i in range(2):
            t = O.transformer_block(t, p, O.block_prefix(variant, name, i), le, biases[rwi] if rwi else None)
            tr[f"blk{bi[0]}.out"] = t
            bi[0] += 1
        return t

    t = y.permute(0, 2, 1)
    x1 = O.patch_merge(stage(t, "dtransformer1", 1), p, "pm1"); tr["p1"] = x1
    x2 = O.patch_merge(stage(x1, "dtransformer2", 2), p, "pm2"); tr["p2"] = x2
    x3 = O.patch_merge(stage(x2, "dtransformer3", 3), p, "pm3"); tr["p3"] = x3
    x4 = O.patch_merge(stage(x3, "dtransformer34", 4), p, "pm4"); tr["p4"] = x4
    xm = stage(x4, "transformer", 0) + x4; tr["xmid"] = xm
    d = O.patch_separate(stage(xm, "utransformer4", 0), p, "ps4") + x3; tr["u3"] = d
    d = O.patch_separate(stage(d, "utranformer3", 4), p, "ps3") + x2; tr["u2"] = d
    d = O.patch_separate(stage(d, "utransformer2", 3), p, "ps2") + x1; tr["u1"] = d
    d = O.patch_separate(stage(d, "utransformer1", 2), p, "ps1"); tr["u0"] = d
    d = d.permute(0, 2, 1) + x0
    out = F.conv1d(d, p["transconv.0.weight"], p["transconv.0.bias"], padding=1)
    return out, tr


def make_case(variant, leads, L, B, seed=1234, dtype=torch.float64):
    p32 = O.init_params(O.ralenet_param_shapes(variant, leads), seed)
    g = torch.Generator().manual_seed(2023)
    x = torch.randn(B, leads, L, generator=g)
    tgt = torch.randn(B, leads, L, generator=g)
    return p32, x, tgt


def run_parity(variant, leads, L, B, device="cuda:0", trace=True, seed=1234):
    """-> dict of relative errors (HIP fp32 vs fp64 oracle)."""
    from ecg_denoise_amd import RALENet
    p32, x, tgt = make_case(variant, leads, L, B, seed)
    model = RALENet(variant, leads=leads, L=L, max_batch=B, train=True, device=device)
    model.load_state_dict(p32, strict=False)
    xd, td = x.to(device), tgt.to(device)
    model.train()
    y = model(xd)
    loss, snr, rmse = model.loss_and_metrics(y, td)
    model.backward()
    torch.cuda.synchronize()
    # fp64 oracle
    p = OrderedDict((k, v.double().requires_grad_(True)) for k, v in p32.items())
    bn = O.new_bn_state(8, torch.float64)
    yo, tr = oracle_trace(p, x.double(), variant, bn)
    lo = O.mse(yo, tgt.double())
    grads = torch.autograd.grad(lo, list(p.values()), allow_unused=True)
    res = OrderedDict()
    res["y"] = rel(y.cpu().numpy(), yo.detach().numpy())
    res["loss"] = abs(loss.item() - lo.item()) / abs(lo.item())
    res["snr"] = rel(snr.cpu().numpy(), O.snr(tgt.double(), yo.detach()).numpy())
    res["rmse"] = rel(rmse.cpu().numpy(), O.rmse(tgt.double(), yo.detach()).numpy())
    st = model.state_dict()
    res["bn_mean"] = rel(st["conv1.2.running_mean"].cpu().numpy(), bn["running_mean"].numpy())
    res["bn_var"] = rel(st["conv1.2.running_var"].cpu().numpy(), bn["running_var"].numpy())
    if trace:
        for k, v in tr.items():
            res["act:" + k] = rel(model.debug_tensor(k).cpu().numpy()[:v.numel()], v.detach().numpy())
    ng = model.named_grads()
    gerr = OrderedDict()
    for (k, _), g in zip(p.items(), grads):
        g = torch.zeros_like(p[k]) if g is None else g
        gn = g.norm().item()
        mine = ng[k].cpu().numpy()
        if k.endswith("to_kv.bias"):
            # key-bias gradient is identically zero (softmax shift invariance): compare the value half,
            # and bound the key half by rounding noise
            C = mine.size // 2
            gerr["grad:" + k + "[v]"] = rel(mine[C:], g.numpy()[C:])
            gerr["gradabs:" + k + "[k]"] = float(np.abs(mine[:C]).max())
        else:
            gerr["grad:" + k] = rel(mine, g.numpy()) if gn > 1e-12 else float(np.abs(mine).max())
    res.update(gerr)
    return res, model, (p32, x, tgt)
