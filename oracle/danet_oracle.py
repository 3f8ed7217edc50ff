"""CPU restatement of the reference's DANet comparison baseline, `model/DAM.py::Seq2Seq2` (TEST INFRASTRUCTURE ONLY).

  DeNoiseEnc (DAM.py:80-98)    4 x EncoderCell (:51-77): Conv1d(stride 2; k 17,17,3,3; pad 8,8,1,1; 2->4->8->16->32)
                               -> APReLU -> BatchNorm1d
  APReLU (:12-48)              p = max(x, 0), n = min(x, 0); [mean_L p, mean_L n] (B, 2C) -> Linear(2C, 2C) -> BatchNorm1d
                               (over the batch) -> ReLU -> Linear(2C, C) -> BatchNorm1d -> Sigmoid = alpha; p + alpha * n
  DeNoiseDec (:310-338)        4 x DecoderCell (:158-190): ConvTranspose1d(stride 2; k 4,4,18,18; pad 1,1,8,8;
                               32->16->8->4->2) -> APReLU -> BatchNorm1d -> DAM (not in the last cell); the input of cells
                               1..3 is alignment_add(previous output, encoder feature) (:283-308; equal lengths whenever L
                               is a multiple of 16, which is all this build accepts)
  DAM (:101-155)               channel attention sigmoid(fcn(mean_L x) + fcn(max_L x)) with ONE fcn applied twice (fcn1 and
                               fcn2 are built from the same module list: shared weights, and in training its two
                               BatchNorm1d see - and update their running statistics with - two batches per forward);
                               spatial attention sigmoid(conv1x1([mean_C x; max_C x])), computed from the input x;
                               out = Sattn * (Cattn * x)

Functional, dtype-generic (fp64 for the parity tests).  `state` holds running_mean / running_var per BatchNorm, updated in
place in training mode exactly as torch does (momentum 0.1, unbiased variance).  Pinned against the reference itself by
tests/golden/g3_danet_L512.npz (oracle/gen_golden_danet.py imports model/DAM.py)."""
from collections import OrderedDict
import math

import numpy as np
import torch
import torch.nn.functional as F

ENC_CH, ENC_K, ENC_P = [4, 8, 16, 32], [17, 17, 3, 3], [8, 8, 1, 1]
DEC_CH, DEC_K, DEC_P = [16, 8, 4, 2], [4, 4, 18, 18], [1, 1, 8, 8]


def _bn_keys(d, pre, c):
    d[pre + ".weight"] = (c,); d[pre + ".bias"] = (c,)
    d[pre + ".running_mean"] = (c,); d[pre + ".running_var"] = (c,); d[pre + ".num_batches_tracked"] = ()


def _aprelu_keys(d, pre, c):
    d[pre + ".fcn.0.weight"] = (2 * c, 2 * c); d[pre + ".fcn.0.bias"] = (2 * c,)
    _bn_keys(d, pre + ".fcn.1", 2 * c)
    d[pre + ".fcn.3.weight"] = (c, 2 * c); d[pre + ".fcn.3.bias"] = (c,)
    _bn_keys(d, pre + ".fcn.4", c)


def danet_state_shapes(leads=2):
    """every state_dict entry of Seq2Seq2 in the reference's order (fcn2.* are the same tensors as fcn1.*)"""
    d = OrderedDict()
    cin = leads
    for i in range(4):
        pre = f"enc.EncoderList.cell{i}"
        d[pre + ".conv.weight"] = (ENC_CH[i], cin, ENC_K[i]); d[pre + ".conv.bias"] = (ENC_CH[i],)
        _aprelu_keys(d, pre + ".activate", ENC_CH[i])
        _bn_keys(d, pre + ".bn", ENC_CH[i])
        cin = ENC_CH[i]
    for i in range(4):
        pre = f"dec.DecoderList.{i}"
        c = DEC_CH[i]
        d[pre + ".deconv.weight"] = (cin, c, DEC_K[i]); d[pre + ".deconv.bias"] = (c,)
        _aprelu_keys(d, pre + ".activate", c)
        _bn_keys(d, pre + ".bn", c)
        if i < 3:
            for f in ("fcn1", "fcn2"):
                d[f"{pre}.dam.{f}.0.weight"] = (c, c); d[f"{pre}.dam.{f}.0.bias"] = (c,)
                _bn_keys(d, f"{pre}.dam.{f}.1", c)
                d[f"{pre}.dam.{f}.3.weight"] = (c, c); d[f"{pre}.dam.{f}.3.bias"] = (c,)
                _bn_keys(d, f"{pre}.dam.{f}.4", c)
            d[pre + ".dam.convsa.weight"] = (1, 2, 1); d[pre + ".dam.convsa.bias"] = (1,)
        cin = c
    return d


def is_param(k):
    return not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))


def init_state(seed, leads=2, dtype=torch.float32):
    """build-owned deterministic values for every entry: PyTorch-default scale for weights and biases, BatchNorm affines
    1/0 + 0.1 N(0,1), running statistics away from their 0/1 defaults (so that the eval path is really tested);
    fcn2.* alias fcn1.*"""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    fan = None
    for k, shp in danet_state_shapes(leads).items():
        if ".dam.fcn2." in k:
            out[k] = out[k.replace(".dam.fcn2.", ".dam.fcn1.")]
            continue
        if k.endswith("num_batches_tracked"):
            out[k] = torch.tensor(0, dtype=torch.int64); continue
        if k.endswith("running_mean"):
            a = 0.2 * rng.standard_normal(shp)
        elif k.endswith("running_var"):
            a = 1.0 + 0.5 * rng.random(shp)
        elif len(shp) >= 2:
            fan = int(np.prod(shp[1:])) if "deconv" not in k else shp[0] * shp[2]
            a = rng.uniform(-1.0 / math.sqrt(fan), 1.0 / math.sqrt(fan), shp)
        elif ".bn." in k or ".fcn.1." in k or ".fcn.4." in k or ".fcn1.1." in k or ".fcn1.4." in k:
            a = (1.0 if k.endswith("weight") else 0.0) + 0.1 * rng.standard_normal(shp)
        else:
            a = rng.uniform(-1.0 / math.sqrt(fan), 1.0 / math.sqrt(fan), shp)
        out[k] = torch.tensor(a, dtype=dtype)
    return out


def _bn(x, st, pre, training, eps=1e-5, momentum=0.1):
    """BatchNorm1d over (B, C) or (B, C, L) with torch's running-statistic update (in place on st)"""
    w, b = st[pre + ".weight"], st[pre + ".bias"]
    dims = (0,) if x.dim() == 2 else (0, 2)
    shape = (1, -1) if x.dim() == 2 else (1, -1, 1)
    if training:
        mean = x.mean(dims); var = x.var(dims, unbiased=False)
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            st[pre + ".running_mean"].mul_(1 - momentum).add_(momentum * mean.detach().to(st[pre + ".running_mean"].dtype))
            st[pre + ".running_var"].mul_(1 - momentum).add_(momentum * (var.detach() * n / max(n - 1, 1)).to(st[pre + ".running_var"].dtype))
            st[pre + ".num_batches_tracked"] += 1
    else:
        mean, var = st[pre + ".running_mean"].to(x.dtype), st[pre + ".running_var"].to(x.dtype)
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + eps) * w.view(shape) + b.view(shape)


def _aprelu(x, st, pre, training):
    p, n = torch.clamp(x, min=0), torch.clamp(x, max=0)
    d = torch.cat([p.mean(-1), n.mean(-1)], 1)
    h = F.linear(d, st[pre + ".fcn.0.weight"], st[pre + ".fcn.0.bias"])
    h = torch.relu(_bn(h, st, pre + ".fcn.1", training))
    h = F.linear(h, st[pre + ".fcn.3.weight"], st[pre + ".fcn.3.bias"])
    alpha = torch.sigmoid(_bn(h, st, pre + ".fcn.4", training))
    return p + alpha.unsqueeze(2) * n


def _dam_fcn(v, st, pre, training):
    h = F.linear(v, st[pre + ".0.weight"], st[pre + ".0.bias"])
    h = torch.relu(_bn(h, st, pre + ".1", training))
    h = F.linear(h, st[pre + ".3.weight"], st[pre + ".3.bias"])
    return torch.sigmoid(_bn(h, st, pre + ".4", training))


def _dam(x, st, pre, training):
    ga = _dam_fcn(x.mean(-1), st, pre + ".fcn1", training)          # the shared fcn: first the average-pooled batch,
    gm = _dam_fcn(x.amax(-1), st, pre + ".fcn1", training)          # then the max-pooled one
    cattn = torch.sigmoid(ga + gm).unsqueeze(-1)
    cat = torch.stack([x.mean(1), x.amax(1)], 1)                    # (B, 2, L)
    sattn = torch.sigmoid(F.conv1d(cat, st[pre + ".convsa.weight"], st[pre + ".convsa.bias"]))   # (B, 1, L)
    return sattn * (cattn * x)


def danet_forward(st, x, training):
    """Seq2Seq2.forward (DAM.py:341-349) on x (B, leads, L), L a multiple of 16"""
    feats = []
    for i in range(4):
        pre = f"enc.EncoderList.cell{i}"
        x = F.conv1d(x, st[pre + ".conv.weight"], st[pre + ".conv.bias"], stride=2, padding=ENC_P[i])
        x = _bn(_aprelu(x, st, pre + ".activate", training), st, pre + ".bn", training)
        feats.append(x)
    for i in range(4):
        pre = f"dec.DecoderList.{i}"
        if i > 0:
            x = x + feats[3 - i]
        x = F.conv_transpose1d(x, st[pre + ".deconv.weight"], st[pre + ".deconv.bias"], stride=2, padding=DEC_P[i])
        x = _bn(_aprelu(x, st, pre + ".activate", training), st, pre + ".bn", training)
        if i < 3:
            x = _dam(x, st, pre + ".dam", training)
    return x
