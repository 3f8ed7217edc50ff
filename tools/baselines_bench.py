import os
"""Throughput of the comparison baselines (SURVEY 8 f4) on one MI355X at batch 2048 x 2 x 512: train step and inference
forward of ACDAE and DANet, and the db8 wavelet-threshold denoiser.  One JSON line per model."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecg_denoise_amd import ACDAE, DANet, wavelet_denoise
B, L = int(os.environ.get("B", 2048)), 512
g = torch.Generator().manual_seed(2023)
x = torch.randn(B, 2, L, generator=g).cuda(); t = torch.randn(B, 2, L, generator=g).cuda()
def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for name, cls, extra in (("acdae", ACDAE, {"fwd_GFLOP_per_step": round(B * 2 * 9.87e6 / 1e9, 1)}),
                         ("danet", DANet, {"launches_per_step": 38 + 38 + 2})):
    m = cls(L=L, max_batch=B, device="cuda:0", seed=1)
    m.train(); tt = timeit(lambda: m.train_step(x, t))
    m.eval(); ti = timeit(lambda: m(x))
    print(json.dumps({"model": name, "batch": B, "train_ms": round(tt * 1e3, 3), "train_windows_per_s": round(B / tt),
                      "infer_ms": round(ti * 1e3, 3), "infer_windows_per_s": round(B / ti), **extra}))
    del m
tw = timeit(lambda: wavelet_denoise(x))
print(json.dumps({"model": "wavelet db8 threshold", "batch": B, "infer_ms": round(tw * 1e3, 4), "infer_windows_per_s": round(B / tw),
                  "hbm_GBps": round(2 * x.numel() * 4 / tw / 1e9, 1)}))
