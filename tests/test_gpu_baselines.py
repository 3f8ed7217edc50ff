"""db8 wavelet-threshold baseline on the GPU (ral_wavelet_denoise, through the C ABI) against the fp64 oracle
(oracle/baselines_oracle.py, the reference's denoisefunc.py:7-33 protocol) and against the reference function's own outputs
on PyWavelets (tests/golden/g8_wavelet.npz).  fp32 tolerance 1e-5 relative L2."""
import os

import numpy as np
import pytest
import torch

import baselines_oracle as W

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("shape", [(5, 2, 512), (7, 256), (3, 1, 1024), (4, 2, 30), (2, 2, 1000), (3, 4096), (2, 14)])
def test_wavelet_denoise_matches_oracle(shape):
    from ecg_denoise_amd import wavelet_denoise
    g = np.random.default_rng(sum(shape))
    # an ECG-like record: slow wave + spikes + noise (white noise alone leaves nothing above the threshold to compare)
    t = np.arange(shape[-1])
    x = (np.sin(2 * np.pi * t / 97.0) + 3.0 * (t % 181 == 90) + 0.3 * g.standard_normal(shape)).astype(np.float32)
    ref = W.wavelet_denoise(x.astype(np.float64))
    y = wavelet_denoise(x)                              # NumPy in -> NumPy out, like the reference function
    assert isinstance(y, np.ndarray) and y.shape == x.shape and y.dtype == np.float32
    assert _rel(y.astype(np.float64), ref) < 1e-5
    yt = wavelet_denoise(torch.from_numpy(x).cuda())    # device tensor in -> device tensor out
    assert yt.is_cuda and np.array_equal(yt.cpu().numpy(), y)


def test_wavelet_denoise_properties_at_the_bench_batch():
    """BASELINE batch (2048 x 2 x 512): zero threshold is the identity to fp32 rounding, the operator is positively
    homogeneous, in-place use is allowed, and 16 sampled rows match the oracle."""
    import ctypes as C
    from ecg_denoise_amd import _lib, wavelet_denoise
    g = torch.Generator().manual_seed(2023)
    x = torch.randn(2048, 2, 512, generator=g).cuda()
    y0 = wavelet_denoise(x, threshold=0.0)
    assert ((y0 - x).norm() / x.norm()).item() < 2e-6
    y = wavelet_denoise(x)
    y4 = wavelet_denoise(4.0 * x)
    assert torch.equal(y4, 4.0 * y)                     # power-of-two scale: exact in floating point
    rows = [0, 1, 77, 1024, 2047]
    ref = W.wavelet_denoise(x[rows].cpu().numpy().astype(np.float64))
    assert _rel(y[rows].cpu().numpy().astype(np.float64), ref) < 1e-5
    z = x.clone()
    _lib.check(_lib.lib().ral_wavelet_denoise(C.c_void_p(z.data_ptr()), C.c_void_p(z.data_ptr()), 4096, 512, 0.04,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(z, y)


def test_wavelet_denoise_rejects_bad_arguments():
    from ecg_denoise_amd import RalError, wavelet_denoise
    with pytest.raises(ValueError):
        wavelet_denoise(np.zeros((2, 511), np.float32))
    with pytest.raises(ValueError):
        wavelet_denoise(np.zeros(512, np.float32))
    with pytest.raises(RalError):
        wavelet_denoise(np.zeros((2, 512), np.float32), threshold=-1.0)


def test_wavelet_denoise_matches_the_reference_function_golden(golden_dir):
    """the reference's `wavelet_denoise` on PyWavelets 1.1.1 (oracle/gen_golden_wavelet.py): 2-D inputs of three even
    lengths, the 3-D input and the float32 input (the kernel takes even record lengths)"""
    from ecg_denoise_amd import wavelet_denoise
    g = np.load(os.path.join(golden_dir, "g8_wavelet.npz"))
    for kx, ky in (("x_256", "y_256"), ("x_300", "y_300"), ("x_512", "y_512"), ("x_1024", "y_1024"), ("x3_512", "y3_512"),
                   ("x32_512", "y32_512")):
        y = wavelet_denoise(g[kx].astype(np.float32))
        assert y.shape == g[ky].shape
        assert _rel(y.astype(np.float64), g[ky].astype(np.float64)) < 1e-5, kx
