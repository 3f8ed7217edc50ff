"""CPU oracle for the RA-LENet / U-Net hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional torch-CPU ops (own code, autograd for the
backward), the arithmetic that the reference performs on its hot path.  It is the
checker for the HIP kernels: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The product path
(`ecg_denoise_amd/`) never imports it and has no CPU fallback.

Parity pin: the reference has no tests or golden vectors of its own, so
the oracle is pinned against the *reference itself*, imported in the build container
by `oracle/gen_golden.py`, which writes the fixtures under `tests/golden/`
(`tests/test_oracle_golden.py` replays them without the reference).

Reference lines followed (relative to /root/reference):
  * TransformerBlock      model/raletransformer.py:383-410, model/transformer.py:383-411
  * AbsPositionalEncoding model/raletransformer.py:165-183
  * LinearProjection      model/raletransformer.py:239-249
  * MSAttention           model/raletransformer.py:291-322 (no mask), model/transformer.py:289-323 (mask)
  * Mlp / PartialConv_1d  model/raletransformer.py:147-160, 15-58
  * PatchMerging/Separate model/raletransformer.py:411-459
  * RelativePositionEmbedding / mask_fill  model/transformer.py:508-558
  * ralenet.forward       model/raletransformer.py:639-680, model/transformer.py:621-667
  * UNet                  model/UNet.py:46-141
  * newrale               model/ralenet_12leads.py:680-709
  * SNR / RMSE            local_utils/evaluate.py:10-51
  * train step            denoise_train.py:47-59 (Adam lr 1e-3, mse mean)
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

CHANNELS = [8, 16, 32, 64, 128]
RW_LEN = [32, 16, 8, 4]
VARIANTS = ("nra", "full", "mlp")

# (module name, level, index of the R-wave table used (1-based) or 0)
BLOCK_STAGES = [
    ("dtransformer1", 0, 1), ("dtransformer2", 1, 2), ("dtransformer3", 2, 3),
    ("dtransformer34", 3, 4), ("transformer", 4, 0), ("utransformer4", 4, 0),
    ("utranformer3", 3, 4), ("utransformer2", 2, 3), ("utransformer1", 1, 2),
]


def variant_flags(variant):
    """-> (local_enhancement, rwave_bias, basiclayer_naming); main.py:69-77."""
    if variant == "nra":
        return True, False, False
    if variant == "full":
        return True, True, True
    if variant == "mlp":
        return False, True, True
    raise ValueError(variant)


def block_prefix(variant, stage, i):
    return f"{stage}.blocks.{i}." if variant_flags(variant)[2] else f"{stage}.{i}."


# --------------------------------------------------------------------------- params
def ralenet_param_shapes(variant="full", leads=2):
    """Ordered {state_dict key: shape} of trainable parameters (reference order)."""
    le, rw, _ = variant_flags(variant)
    d = OrderedDict()
    d["conv1.0.weight"] = (8, leads, 3)
    d["conv1.0.bias"] = (8,)
    d["conv1.2.weight"] = (8,)
    d["conv1.2.bias"] = (8,)
    if rw:
        for i, ln in enumerate(RW_LEN):
            d[f"rwattn{i+1}.relative_position_bias_table"] = (2 * ln - 1, CHANNELS[i] // 4)

    def block(prefix, C):
        d[prefix + "attn.qkv_proj.to_q.weight"] = (C, C)
        d[prefix + "attn.qkv_proj.to_q.bias"] = (C,)
        d[prefix + "attn.qkv_proj.to_kv.weight"] = (2 * C, C)
        d[prefix + "attn.qkv_proj.to_kv.bias"] = (2 * C,)
        d[prefix + "attn.proj.weight"] = (C, C)
        d[prefix + "attn.proj.bias"] = (C,)
        d[prefix + "norm1.weight"] = (C,)
        d[prefix + "norm1.bias"] = (C,)
        d[prefix + "norm2.weight"] = (C,)
        d[prefix + "norm2.bias"] = (C,)
        d[prefix + "mlp.fc1.weight"] = (4 * C, C)
        d[prefix + "mlp.fc1.bias"] = (4 * C,)
        d[prefix + "mlp.fc2.weight"] = (C, 4 * C)
        d[prefix + "mlp.fc2.bias"] = (C,)
        if le:
            d[prefix + "mlp.leconv.partial_conv3.weight"] = (1, 1, 3)

    def stage(name, lvl):
        for i in range(2):
            block(block_prefix(variant, name, i), CHANNELS[lvl])

    def pm(name, C):  # PatchMerging(dim=C): Linear(2C,2C,no bias)+LN(2C)
        d[name + ".reduction.weight"] = (2 * C, 2 * C)
        d[name + ".norm.weight"] = (2 * C,)
        d[name + ".norm.bias"] = (2 * C,)

    def ps(name, C):  # PatchSeparate(dim=C): Linear(C/2,C/2,no bias)+LN(C/2)
        d[name + ".reduction.weight"] = (C // 2, C // 2)
        d[name + ".norm.weight"] = (C // 2,)
        d[name + ".norm.bias"] = (C // 2,)

    stage("dtransformer1", 0); pm("pm1", 8)
    stage("dtransformer2", 1); pm("pm2", 16)
    stage("dtransformer3", 2); pm("pm3", 32)
    stage("dtransformer34", 3); pm("pm4", 64)
    stage("transformer", 4)
    stage("utransformer4", 4); ps("ps4", 128)
    stage("utranformer3", 3); ps("ps3", 64)
    stage("utransformer2", 2); ps("ps2", 32)
    stage("utransformer1", 1); ps("ps1", 16)
    d["transconv.0.weight"] = (leads, 8, 3)
    d["transconv.0.bias"] = (leads,)
    return d


def init_params(shapes, seed, dtype=torch.float32):
    """Build-owned deterministic init rule: one numpy Generator seeded
    with `seed`, keys visited in order.  Linear/Conv weights and the bias that follows
    them ~U(-1/sqrt(fan_in), 1/sqrt(fan_in)) with fan_in = prod(shape[1:]) (PyTorch's
    default scale); norm affines are 1/0 perturbed by 0.1*N(0,1) and the R-wave tables
    are 0.02*N(0,1) (the reference starts them at exactly 1/0/0, transformer.py:514,
    which would leave those code paths untested)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    fan = None
    for k, shp in shapes.items():
        if "relative_position_bias_table" in k:
            a = rng.standard_normal(shp) * 0.02
        elif len(shp) == 1 and (".norm" in k or ".bn." in k or k.startswith("conv1.2.")
                                or k.startswith("bottleneck.2.") or k.startswith("bottleneck.5.")):
            a = (1.0 if k.endswith("weight") else 0.0) + 0.1 * rng.standard_normal(shp)
        elif len(shp) >= 2:
            fan = int(np.prod(shp[1:]))
            a = rng.uniform(-1.0 / math.sqrt(fan), 1.0 / math.sqrt(fan), shp)
        else:
            b = 1.0 / math.sqrt(fan) if fan else 0.1
            a = rng.uniform(-b, b, shp)
        out[k] = torch.tensor(a, dtype=dtype)
    return out


# --------------------------------------------------------------------------- blocks
def pe_table(n, C, dtype=torch.float32):
    """Sinusoid table, fp32 arithmetic in the reference's op order
    (raletransformer.py:172-181); cast afterwards."""
    X = torch.arange(n, dtype=torch.float32).reshape(-1, 1) / torch.pow(
        10000, torch.arange(0, C, 2, dtype=torch.float32) / C)
    P = torch.zeros(n, C, dtype=torch.float32)
    P[:, 0::2] = torch.sin(X)
    P[:, 1::2] = torch.cos(X)
    return P.to(dtype)


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def rwave_bias(table, Len, W):
    """(2Len-1, h) table -> (h, W, W) additive logits bias, zero outside the centred
    Len x Len window (transformer.py:534-558)."""
    idx = torch.arange(Len)
    rel = idx[:, None] - idx[None, :] + Len - 1          # (Len, Len)
    b = table[rel.reshape(-1)].reshape(Len, Len, -1).permute(2, 0, 1)
    off = (W - Len) // 2
    return F.pad(b, (off, W - Len - off, off, W - Len - off))


def transformer_block(x, p, pre, le, bias=None):
    """x (B,N,C) -> (B,N,C); SURVEY §3.3."""
    B, N, C = x.shape
    h = C // 4
    t = x * math.sqrt(C) + pe_table(N, C, x.dtype)
    t = F.layer_norm(t, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-5)
    q = F.linear(t, p[pre + "attn.qkv_proj.to_q.weight"], p[pre + "attn.qkv_proj.to_q.bias"])
    kv = F.linear(t, p[pre + "attn.qkv_proj.to_kv.weight"], p[pre + "attn.qkv_proj.to_kv.bias"])
    q = q.reshape(B, N, h, 4).permute(0, 2, 1, 3) * 0.5           # head_dim**-0.5
    k = kv[..., :C].reshape(B, N, h, 4).permute(0, 2, 1, 3)
    v = kv[..., C:].reshape(B, N, h, 4).permute(0, 2, 1, 3)
    s = q @ k.transpose(-1, -2)
    if bias is not None:
        s = s + bias.unsqueeze(0)
    a = torch.softmax(s, dim=-1)
    o = (a @ v).permute(0, 2, 1, 3).reshape(B, N, C)
    x = x + F.linear(o, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])
    g = F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-5)
    u = gelu(F.linear(g, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"]))
    if le:
        w = p[pre + "mlp.leconv.partial_conv3.weight"].reshape(1, 1, 3)
        c0 = F.conv1d(u[:, :, 0].unsqueeze(1), w, padding=1).squeeze(1)   # over tokens
        u = gelu(torch.cat([c0.unsqueeze(-1), u[:, :, 1:]], dim=-1))
    return x + F.linear(u, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])


def patch_merge(x, p, name):
    B, N, C = x.shape
    x = x.reshape(B, N // 2, 2 * C)      # == cat(x[:,0::2], x[:,1::2], -1)
    x = F.layer_norm(x, (2 * C,), p[name + ".norm.weight"], p[name + ".norm.bias"], 1e-5)
    return F.linear(x, p[name + ".reduction.weight"])


def patch_separate(x, p, name):
    B, N, C = x.shape
    x = torch.cat([x[:, :, :C // 2], x[:, :, C // 2:]], dim=1)   # 'b l (c1 c2) -> b (c1 l) c2'
    x = F.layer_norm(x, (C // 2,), p[name + ".norm.weight"], p[name + ".norm.bias"], 1e-5)
    return F.linear(x, p[name + ".reduction.weight"])


def new_bn_state(n=8, dtype=torch.float32):
    return {"running_mean": torch.zeros(n, dtype=dtype), "running_var": torch.ones(n, dtype=dtype),
            "num_batches_tracked": 0}


def ralenet_forward(p, x, variant="full", training=True, bn=None):
    """x (B, leads, L) -> (B, leads, L).  `bn` is updated in place when training."""
    le, rw, _ = variant_flags(variant)
    B, _, L = x.shape
    if bn is None:
        bn = new_bn_state(8, x.dtype)
    y = F.conv1d(x, p["conv1.0.weight"], p["conv1.0.bias"], padding=1)
    y = F.leaky_relu(y, 0.2)
    y = F.batch_norm(y, bn["running_mean"], bn["running_var"], p["conv1.2.weight"],
                     p["conv1.2.bias"], training, 0.1, 1e-5)
    if training:
        bn["num_batches_tracked"] += 1
    x0 = y
    biases = [None] * 5
    if rw:
        for i, ln in enumerate(RW_LEN):
            biases[i + 1] = rwave_bias(p[f"rwattn{i+1}.relative_position_bias_table"], ln, L >> i)

    def stage(t, name, rwi):
        for i in range(2):
            t = transformer_block(t, p, block_prefix(variant, name, i), le, biases[rwi] if rwi else None)
        return t

    t = y.permute(0, 2, 1)
    x1 = patch_merge(stage(t, "dtransformer1", 1), p, "pm1")
    x2 = patch_merge(stage(x1, "dtransformer2", 2), p, "pm2")
    x3 = patch_merge(stage(x2, "dtransformer3", 3), p, "pm3")
    x4 = patch_merge(stage(x3, "dtransformer34", 4), p, "pm4")
    xm = stage(x4, "transformer", 0) + x4
    d = patch_separate(stage(xm, "utransformer4", 0), p, "ps4") + x3
    d = patch_separate(stage(d, "utranformer3", 4), p, "ps3") + x2
    d = patch_separate(stage(d, "utransformer2", 3), p, "ps2") + x1
    d = patch_separate(stage(d, "utransformer1", 2), p, "ps1")
    d = d.permute(0, 2, 1) + x0
    return F.conv1d(d, p["transconv.0.weight"], p["transconv.0.bias"], padding=1)


# --------------------------------------------------------------------------- UNet
UNET_CH = [2, 4, 8, 16, 32]


def unet_param_shapes(leads=2):
    d = OrderedDict()
    ch = [leads] + UNET_CH[1:]
    for i in range(4):
        d[f"EncList.{i}.conv.weight"] = (ch[i + 1], ch[i], 3)
        d[f"EncList.{i}.conv.bias"] = (ch[i + 1],)
        d[f"EncList.{i}.bn.weight"] = (ch[i + 1],)
        d[f"EncList.{i}.bn.bias"] = (ch[i + 1],)
    for i in range(4):
        cin, cout = ch[4 - i], ch[3 - i]
        d[f"DecList.{i}.conv.weight"] = (cin, cout, 4)
        d[f"DecList.{i}.conv.bias"] = (cout,)
        d[f"DecList.{i}.bn.weight"] = (cout,)
        d[f"DecList.{i}.bn.bias"] = (cout,)
    d["bottleneck.0.weight"] = (32, 32, 1); d["bottleneck.0.bias"] = (32,)
    d["bottleneck.2.weight"] = (32,); d["bottleneck.2.bias"] = (32,)
    d["bottleneck.3.weight"] = (32, 32, 3); d["bottleneck.3.bias"] = (32,)
    d["bottleneck.5.weight"] = (32,); d["bottleneck.5.bias"] = (32,)
    d["bottleneck.6.weight"] = (32, 32, 1); d["bottleneck.6.bias"] = (32,)
    return d


UNET_BN = [f"EncList.{i}.bn" for i in range(4)] + [f"DecList.{i}.bn" for i in range(4)] + \
          ["bottleneck.2", "bottleneck.5"]


def unet_bn_state(p, dtype=torch.float32):
    return {k: new_bn_state(p[k + ".weight"].numel(), dtype) for k in UNET_BN}


def unet_forward(p, x, training=True, bn=None):
    """UNet.py:126-141; LeakyReLU slope 0.01, BN after every conv (Lazy -> eager)."""
    if bn is None:
        bn = unet_bn_state(p, x.dtype)

    def bnf(t, k):
        s = bn[k]
        t = F.batch_norm(t, s["running_mean"], s["running_var"], p[k + ".weight"], p[k + ".bias"],
                         training, 0.1, 1e-5)
        if training:
            s["num_batches_tracked"] += 1
        return t

    feats = []
    for i in range(4):
        x = F.conv1d(x, p[f"EncList.{i}.conv.weight"], p[f"EncList.{i}.conv.bias"], stride=2, padding=1)
        x = F.leaky_relu(bnf(x, f"EncList.{i}.bn"), 0.01)
        if i < 3:
            feats.append(x)
    b = F.conv1d(x, p["bottleneck.0.weight"], p["bottleneck.0.bias"])
    b = bnf(F.leaky_relu(b, 0.01), "bottleneck.2")
    b = F.conv1d(b, p["bottleneck.3.weight"], p["bottleneck.3.bias"], padding=1)
    b = bnf(F.leaky_relu(b, 0.01), "bottleneck.5")
    b = F.conv1d(b, p["bottleneck.6.weight"], p["bottleneck.6.bias"])
    x = b + x
    for i in range(4):
        x = F.conv_transpose1d(x, p[f"DecList.{i}.conv.weight"], p[f"DecList.{i}.conv.bias"],
                               stride=2, padding=1)
        x = bnf(x, f"DecList.{i}.bn")
        if i < 3:
            x = F.leaky_relu(x, 0.01) + feats[2 - i]
    return x


# --------------------------------------------------------------------------- ACDAE (comparison baseline)
ACDAE_CH = [2, 16, 32, 64, 128]
ACDAE_KS = [13, 7, 7, 7]


def acdae_param_shapes():
    """model/ACDAE.py:62-73 in state_dict order: the four encoder convs, then per decoder block the transposed conv
    and the ECA's bias-free 3-tap conv over the channel axis."""
    d = OrderedDict()
    for i in range(4):
        d[f"EncList.{i}.conv.weight"] = (ACDAE_CH[i + 1], ACDAE_CH[i], ACDAE_KS[i])
        d[f"EncList.{i}.conv.bias"] = (ACDAE_CH[i + 1],)
    for i in range(4):
        cin, cout, k = ACDAE_CH[4 - i], ACDAE_CH[3 - i], ACDAE_KS[3 - i]
        d[f"DecList.{i}.conv.weight"] = (cin, cout, k)
        d[f"DecList.{i}.conv.bias"] = (cout,)
        d[f"DecList.{i}.ECA.conv.weight"] = (1, 1, 3)
    return d


def acdae_forward(p, x):
    """model/ACDAE.py:75-86.  EncBlock (:25-39): Conv1d('same') -> MaxPool1d(2) -> LeakyReLU(0.01).  DecBlock (:42-59):
    ConvTranspose1d(stride 1, pad (k-1)/2) -> Upsample(x2, linear, align_corners=False) -> LeakyReLU -> ECA (:10-22:
    channel means -> conv k3 over the channel axis, no bias -> sigmoid -> scale).  Skips are added after the ECA."""
    feats = []
    for i in range(4):
        k = ACDAE_KS[i]
        x = F.conv1d(x, p[f"EncList.{i}.conv.weight"], p[f"EncList.{i}.conv.bias"], padding=(k - 1) // 2)
        x = F.leaky_relu(F.max_pool1d(x, 2), 0.01)
        if i < 3:
            feats.append(x)
    for i in range(4):
        k = ACDAE_KS[3 - i]
        x = F.conv_transpose1d(x, p[f"DecList.{i}.conv.weight"], p[f"DecList.{i}.conv.bias"], padding=(k - 1) // 2)
        x = F.leaky_relu(F.interpolate(x, scale_factor=2, mode="linear", align_corners=False), 0.01)
        m = x.mean(-1, keepdim=True)                                             # (B, C, 1)
        z = F.conv1d(m.transpose(-1, -2), p[f"DecList.{i}.ECA.conv.weight"], padding=1).transpose(-1, -2)
        x = x * torch.sigmoid(z)
        if i < 3:
            x = x + feats[2 - i]
    return x


# --------------------------------------------------------------------------- newrale
def newrale_param_shapes():
    d = OrderedDict()
    d["conv1.weight"] = (6, 12, 13); d["conv1.bias"] = (6,)
    d["conv2.weight"] = (2, 6, 13); d["conv2.bias"] = (2,)
    d["conv3.weight"] = (6, 2, 13); d["conv3.bias"] = (6,)
    d["conv4.weight"] = (12, 6, 13); d["conv4.bias"] = (12,)
    return d


def newrale_forward(pa, p, x, variant="full", training=True, bn=None):
    """ralenet_12leads.py:698-709: 12->6->2 (k13, LReLU 0.01) -> ralenet -> 2->6->12."""
    x = F.leaky_relu(F.conv1d(x, pa["conv1.weight"], pa["conv1.bias"], padding=6), 0.01)
    x = F.leaky_relu(F.conv1d(x, pa["conv2.weight"], pa["conv2.bias"], padding=6), 0.01)
    x = ralenet_forward(p, x, variant, training, bn)
    x = F.leaky_relu(F.conv1d(x, pa["conv3.weight"], pa["conv3.bias"], padding=6), 0.01)
    return F.conv1d(x, pa["conv4.weight"], pa["conv4.bias"], padding=6)


# --------------------------------------------------------------------------- metrics / optimiser
def mse(pred, target):
    return ((pred - target) ** 2).mean()


def snr(y, y_pred):
    """evaluate.py:31-51 -> (B,)"""
    y = y.flatten(1); y_pred = y_pred.flatten(1)
    return 10 * torch.log10((y ** 2).mean(-1) / ((y - y_pred) ** 2).mean(-1))


def rmse(y, y_pred):
    """evaluate.py:10-29 -> (B,)"""
    y = y.flatten(1); y_pred = y_pred.flatten(1)
    return torch.sqrt(((y - y_pred) ** 2).mean(-1))


def adam_step(params, grads, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (denoise_train.py:24); `step` is 1-based.  In place."""
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k in params:
        g = grads[k]
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-lr / bc1)


def train_step(p, x, target, forward, m, v, step, lr=1e-3):
    """One denoise_train.py:51-59 iteration.  -> dict(loss, pred, snr, rmse, grads)."""
    leaves = OrderedDict((k, t.detach().clone().requires_grad_(True)) for k, t in p.items())
    pred = forward(leaves, x)
    loss = mse(pred, target)
    grads_t = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    grads = OrderedDict((k, (g if g is not None else torch.zeros_like(p[k])))
                        for k, g in zip(leaves, grads_t))
    with torch.no_grad():
        adam_step(p, grads, m, v, step, lr)
    return {"loss": loss.detach(), "pred": pred.detach(), "snr": snr(target, pred.detach()),
            "rmse": rmse(target, pred.detach()), "grads": grads}
