"""Run the same backward twice and list the gradient tensors that differ (race detector; GPU box only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ecg_denoise_amd import RALENet, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = RALENet("full", leads=1, L=512, max_batch=B, device="cuda:0", seed=1)
x = torch.randn(B, 1, 512, device="cuda:0")
m.train()
y = m(x)
dy = torch.randn_like(y) / y.numel()
gs = []
for _ in range(3):
    m.backward(dy)
    torch.cuda.synchronize()
    gs.append(m.eng.grads.clone())
base = m.eng.grads.data_ptr()
for k, v in m.named_grads().items():   # views into eng.grads: recover the offsets from the data pointers
    off = (v.data_ptr() - base) // 4; n = v.numel()
    a, b, c = (g[off:off + n] for g in gs)
    d = max((a - b).abs().max().item(), (a - c).abs().max().item())
    s_ = a.abs().max().item()
    if s_ > 0 and d / s_ > 1e-5:
        print(f"{k:60s} n={n:7d} maxdiff/max = {d/s_:.3e}")
print("done")
