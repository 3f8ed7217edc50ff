"""Data-parallel plumbing on one GPU: the split train step of dp.HipEngineAdapter equals the fused one, and the early
gradient bucket (decoder half, all-reduced under the rest of the backward pass) is final when its event fires."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_early_gradient_bucket_is_final_when_its_event_fires():
    from ecg_denoise_amd import RALENet
    from ecg_denoise_amd.dp import HipEngineAdapter
    B = 64
    m = RALENet("full", leads=2, L=256, max_batch=B, device=DEV, seed=4)
    x = torch.randn(B, 2, 256, device=DEV); t = torch.randn(B, 2, 256, device=DEV)
    m.train()
    e = HipEngineAdapter(m)
    (o0, n0), (o1, n1) = e.grad_buckets()
    assert o0 == 0 and o1 == n0 and o1 + n1 == m.eng.grads.numel() and n1 > 0
    names = [k for k, _ in m.named_parameters()]
    first_dec = next(k for k in names if k.startswith("utransformer4."))
    assert m.named_grads()[first_dec].data_ptr() == m.eng.grads.data_ptr() + 4 * o1      # the split is at the decoder
    for _ in range(3):      # several times: a premature event would show up as a race
        e.forward_begin(x); pred = e.forward_end(B)
        e.loss(pred, t, B)
        e.backward_begin()
        comm = e.bucket_stream()
        e.bucket_wait(1, comm)
        with torch.cuda.stream(comm):
            early = m.eng.grads[o1:o1 + n1].clone()
        e.backward_end(B)
        e.bucket_wait(0, comm)
        with torch.cuda.stream(comm):
            late = m.eng.grads[:o1].clone()
        torch.cuda.synchronize()
        assert early.abs().sum().item() > 0
        assert torch.equal(early, m.eng.grads[o1:o1 + n1])
        assert torch.equal(late, m.eng.grads[:o1])
    # and the split step is the fused step
    g_split = m.eng.grads.clone()
    y = m(x); m.loss_and_metrics(y, t); m.backward()
    torch.testing.assert_close(g_split, m.eng.grads, rtol=1e-4, atol=1e-7)
