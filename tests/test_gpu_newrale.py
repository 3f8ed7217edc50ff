"""12-lead adapter (reference model/ralenet_12leads.py::newrale) vs the golden vectors from the reference."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

import ralenet_oracle as O
from parity_util import rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_newrale_matches_reference_golden(golden_dir):
    from ecg_denoise_amd import NewRALE, RALENet
    g = np.load(os.path.join(golden_dir, "g3_newrale_L256.npz"))
    inner = RALENet("full", leads=2, L=256, max_batch=2, device=DEV)
    inner.load_state_dict(O.init_params(O.ralenet_param_shapes("full", 2), 1234), strict=False)
    m = NewRALE(inner)
    pa = O.init_params(O.newrale_param_shapes(), 77)
    m.load_state_dict(pa)
    x = torch.tensor(g["x"]).to(DEV); tgt = torch.tensor(g["target"]).to(DEV)
    m.train()
    y = m(x)
    assert rel(y.cpu().numpy(), g["y_train"]) < 1e-5
    loss, snr, rmse = m.loss_and_metrics(y, tgt)
    assert abs(loss.item() - g["loss"]) < 1e-5 * g["loss"]
    m.backward()
    for k, gr in m.named_grads().items():
        assert rel(gr.cpu().numpy(), g["grad_" + k]) < 2e-4, k
    assert int(inner.state_dict()["conv1.2.num_batches_tracked"]) == 1      # quirk A16: inner BN keeps training
    sd = m.state_dict()
    assert list(sd)[:5] == ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "rale.conv1.0.weight"]
    before = inner.eng.params.clone()
    m.step()
    assert torch.equal(before, inner.eng.params)                             # frozen inner weights
    assert not torch.equal(m.params, torch.zeros_like(m.params))
