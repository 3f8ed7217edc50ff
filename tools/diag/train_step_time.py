"""ms per `model.train_step` (the fused path: ral_forward_loss_means, ral_backward, ral_adam_step) at the bench shape, for 1 and 2 leads,
   optionally on a second model created after the first was destroyed (what bench.py's `leads2` leg does)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from ecg_denoise_amd import RALENet, _lib
_lib.apply_options(os.environ.get("RAL_TOOL_OPTIONS", ""))
B, L, n = 2048, 512, int(os.environ.get("TS_N", "100"))
def run(leads):
    m = RALENet("full", leads=leads, L=L, max_batch=B, device="cuda:0", seed=1)
    m.train()
    x = torch.randn(B, leads, L, device="cuda:0"); t = torch.randn(B, leads, L, device="cuda:0")
    for _ in range(5): m.train_step(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m.train_step(x, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    del m, x, t
    torch.cuda.empty_cache()
    return dt
for leads in (1, 2, 1, 2):
    print(f"leads {leads}: {run(leads):.3f} ms per train_step", flush=True)
