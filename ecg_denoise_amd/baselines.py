"""Non-neural comparison baseline of the reference's result tables.

`wavelet_denoise` mirrors `local_utils/denoisefunc.py:7-33` (same name, same accepted shapes: a 2-D (rows, L) or 3-D
(batch, leads, L) array of records, result of the same shape): db8 decomposition at the maximum level, soft threshold of
every detail band at 0.04 * max(band), reconstruction - as one HIP kernel (`ral_wavelet_denoise`, a record and its
coefficient pyramid in LDS).  The reference works on NumPy arrays on the host; a NumPy array is accepted here too (it is
copied to the device and back), a CUDA tensor stays on the device.  No CPU fallback."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def wavelet_denoise(ecg_data, threshold=0.04, device="cuda:0"):
    is_np = isinstance(ecg_data, np.ndarray)
    x = torch.as_tensor(ecg_data)
    if x.dim() not in (2, 3):
        raise ValueError("wavelet_denoise takes a 2-D (rows, L) or 3-D (batch, leads, L) array")   # the reference returns None
    L = x.shape[-1]
    if L % 2 or L > 8192:
        raise ValueError(f"record length must be even and <= 8192, got {L}")
    xd = x.to(device=device if not x.is_cuda else x.device, dtype=torch.float32).contiguous()
    y = torch.empty_like(xd)
    rows = xd.numel() // L if L else 0
    with torch.cuda.device(xd.device):
        _lib.check(_lib.lib().ral_wavelet_denoise(C.c_void_p(xd.data_ptr()), C.c_void_p(y.data_ptr()), rows, L, float(threshold),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    if is_np:
        return y.cpu().numpy().astype(ecg_data.dtype if ecg_data.dtype.kind == "f" else np.float64)
    return y
