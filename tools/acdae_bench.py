"""ACDAE baseline throughput on one MI355X (train step and inference forward at batch 2048 x 2 x 512)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecg_denoise_amd import ACDAE
B, L = int(os.environ.get("B", 2048)), 512
m = ACDAE(L=L, max_batch=B, device="cuda:0", seed=1)
g = torch.Generator().manual_seed(2023)
x = torch.randn(B, 2, L, generator=g).cuda(); t = torch.randn(B, 2, L, generator=g).cuda()
def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
m.train(); tt = timeit(lambda: m.train_step(x, t))
m.eval(); ti = timeit(lambda: m(x))
print(json.dumps({"model": "acdae", "batch": B, "train_ms": round(tt * 1e3, 3), "train_windows_per_s": round(B / tt),
                  "infer_ms": round(ti * 1e3, 3), "infer_windows_per_s": round(B / ti),
                  "fwd_GFLOP_per_step": round(B * 2 * 9.87e6 / 1e9, 1)}))
