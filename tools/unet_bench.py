"""U-Net baseline throughput on one MI355X (HBM-bound conv stages; SURVEY §8d: 100 KB fp32 per window forward).
Prints windows/s for the train step and the inference forward and the achieved stage-granular GB/s."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecg_denoise_amd import UNet, _lib
_lib.apply_options(os.environ.get("RAL_TOOL_OPTIONS", ""))   # library switches for this run (diagnostics), e.g. unet_fused=0
B, leads, L = int(os.environ.get("B", 2048)), 2, 512
m = UNet(leads=leads, L=L, max_batch=B, device="cuda:0", seed=1)
g = torch.Generator().manual_seed(2023)
x = torch.randn(B, leads, L, generator=g).cuda(); t = torch.randn(B, leads, L, generator=g).cuda()
def timeit(fn, n=50, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
m.train(); tt = timeit(lambda: m.train_step(x, t))
tf = timeit(lambda: m(x))
m.eval(); ti = timeit(lambda: m(x))
w = leads * L * 4                      # bytes of one stage tensor of one window
fwd_bytes = (15 + 11 + 2) * w          # reads (incl. 3 skips + residual) + writes + output BatchNorm pass
bwd_bytes = 11 * 6 * w                 # per stage: G, z_out, inputs (1-2) read; producer gradients written (approx.)
print(json.dumps({"model": "unet", "batch": B, "train_ms": round(tt * 1e3, 3), "train_windows_per_s": round(B / tt),
                  "train_fwd_ms": round(tf * 1e3, 3), "infer_ms": round(ti * 1e3, 3), "infer_windows_per_s": round(B / ti),
                  "fwd_algorithmic_GBps": round(B * fwd_bytes / tf / 1e9, 1), "fwd_frac_of_8TBps": round(B * fwd_bytes / tf / 8e12, 4),
                  "infer_algorithmic_GBps": round(B * fwd_bytes / ti / 1e9, 1)}))
