"""Build-container only: golden vectors of the windowing / noise-mixing step from the REFERENCE's own functions
(`np_norm`, `Gnoisegen` in local_utils/local_utils.py and the einops `rearrange` of `batch_norm_snr_iter`), imported
from /root/reference with `wfdb` stubbed (imported at module level, used only by the file readers).
Writes tests/golden/g7_dataprep.npz."""
import os, sys, types
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.modules.setdefault("wfdb", types.ModuleType("wfdb"))
sys.path.insert(0, REF)
from local_utils import local_utils as LU  # noqa: E402
import torch  # noqa: E402
from einops import rearrange  # noqa: E402

rng = np.random.default_rng(20231008)
cases = {}
for name, (T, leads, L, snr) in {"a": (256 * 8, 2, 256, 0.0), "b": (512 * 4, 2, 512, -2.0), "c": (256 * 3, 2, 256, 4.0),
                                 "d": (1024 * 2, 1, 1024, 2.0)}.items():
    # ADC-like integer signals (wfdb d_signal, physical=False) with a baseline offset and slow drift per lead
    t = np.arange(T)[:, None]
    sig = (1024 + 200 * np.sin(2 * np.pi * t / 360.0 * (1 + np.arange(leads))) + 40 * rng.standard_normal((T, leads))).astype(np.int64)
    noise = (30 * rng.standard_normal((T, leads)) + 5 * np.sin(2 * np.pi * t / 900.0)).astype(np.int64)
    b = T // L
    data_sig = LU.np_norm(sig, dim=0)
    noisy = LU.Gnoisegen(data_sig, noise, snr)[0]
    noisy_w = rearrange(torch.FloatTensor(noisy), '(b l) c -> b c l', b=b).numpy()
    clean_w = rearrange(torch.FloatTensor(data_sig), '(b l) c -> b c l', b=b).numpy()
    cases.update({f"{name}_sig": sig.astype(np.float32), f"{name}_noise": noise.astype(np.float32), f"{name}_snr": np.float64(snr),
                  f"{name}_L": np.int64(L), f"{name}_noisy": noisy_w, f"{name}_clean": clean_w})
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g7_dataprep.npz"), **cases)
print("wrote g7_dataprep.npz:", {k: v.shape for k, v in cases.items() if hasattr(v, "shape") and v.ndim})
