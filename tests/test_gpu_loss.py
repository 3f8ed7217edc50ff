"""ral_loss_mean: F.mse_loss / SNR / RMSE (denoise_train.py:53,58-59; local_utils/evaluate.py:10-51) with the mean finished on
the device - the last workgroup to arrive scales the sum and puts the accumulator back to zero, so a training step has no
fill kernel in front of the loss and no division behind it."""
import ctypes as C

import pytest
import torch

from ecg_denoise_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


@pytest.mark.parametrize("B,n,gw", [(37, 1024, 37), (2048, 512, 8192), (5, 1022, 5), (1, 4, 3)])
def test_loss_mean_equals_the_host_formula_and_resets_its_scratch(B, n, gw):
    g = torch.Generator().manual_seed(B + n)
    pred = torch.randn(B, n, generator=g).to(DEV); tgt = torch.randn(B, n, generator=g).to(DEV)
    scratch = torch.zeros(2, dtype=torch.float64, device=DEV)
    L = _lib.lib()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref_mse = ((pred.double() - tgt.double()) ** 2).mean(1)
    for rep in range(3):                       # the scratch must come back to zero after every call
        loss = torch.full((1,), float("nan"), dtype=torch.float64, device=DEV)
        dy = torch.empty_like(pred); snr = torch.empty(B, device=DEV); rmse = torch.empty(B, device=DEV)
        _lib.check(L.ral_loss_mean(_vp(pred), _vp(tgt), n, B, gw, _vp(dy), _vp(snr), _vp(rmse), _vp(loss), _vp(scratch), s))
        torch.cuda.synchronize()
        assert abs(loss.item() - ref_mse.sum().item() / gw) <= 2e-7 * ref_mse.sum().item() / gw     # per-window sums are fp32
        assert scratch.view(torch.int64).abs().sum().item() == 0
        assert torch.allclose(dy.double(), 2.0 * (pred.double() - tgt.double()) / (gw * n), rtol=1e-6, atol=0)
        assert torch.allclose(rmse.double(), ref_mse.sqrt(), rtol=1e-6)
        ref_snr = 10 * torch.log10((tgt.double() ** 2).mean(1) / ref_mse)
        assert torch.allclose(snr.double(), ref_snr, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,n,gw", [(37, 1024, 37), (2048, 512, 8192), (5, 1022, 5)])
def test_loss_means_carry_the_metric_means(B, n, gw):
    """ral_loss_means: the batch means of SNR and RMSE (what denoise_train.py:58-64 logs per step) leave the loss kernel
    with the loss - sums of the per-window fp32 values in double, divided by the global window count."""
    g = torch.Generator().manual_seed(3 * B + n)
    pred = torch.randn(B, n, generator=g).to(DEV); tgt = torch.randn(B, n, generator=g).to(DEV)
    scratch = torch.zeros(64, dtype=torch.float64, device=DEV)
    L = _lib.lib()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref_mse = ((pred.double() - tgt.double()) ** 2).mean(1)
    for rep in range(3):
        means = torch.full((3,), float("nan"), dtype=torch.float64, device=DEV)
        dy = torch.empty_like(pred); snr = torch.empty(B, device=DEV); rmse = torch.empty(B, device=DEV)
        _lib.check(L.ral_loss_means(_vp(pred), _vp(tgt), n, B, gw, _vp(dy), _vp(snr), _vp(rmse), _vp(means), _vp(scratch), s))
        torch.cuda.synchronize()
        assert abs(means[0].item() - ref_mse.sum().item() / gw) <= 2e-7 * ref_mse.sum().item() / gw
        assert abs(means[1].item() - snr.double().sum().item() / gw) <= 1e-12 * abs(snr.double().sum().item() / gw) + 1e-300
        assert abs(means[2].item() - rmse.double().sum().item() / gw) <= 1e-12 * rmse.double().sum().item() / gw
        assert scratch.view(torch.int64).abs().sum().item() == 0
        assert torch.allclose(rmse.double(), ref_mse.sqrt(), rtol=1e-6)
