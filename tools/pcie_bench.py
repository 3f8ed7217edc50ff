"""Training-step throughput when every batch arrives in pinned HOST buffers (the reference's `data.cuda()` per batch,
denoise_train.py:48-49) instead of being resident in HBM: the PCIe-inclusive number DESIGN.md section 6 quotes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecg_denoise_amd import RALENet
B, L = 2048, 512
m = RALENet("full", leads=1, L=L, max_batch=B, device="cuda:0", seed=2023); m.train()
hx = torch.randn(B, 1, L).pin_memory(); ht = torch.randn(B, 1, L).pin_memory()
dx = hx.cuda(); dt_ = ht.cuda()
def run(host, n=20):
    for _ in range(3):
        m.train_step(hx.cuda(non_blocking=True) if host else dx, ht.cuda(non_blocking=True) if host else dt_)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        m.train_step(hx.cuda(non_blocking=True) if host else dx, ht.cuda(non_blocking=True) if host else dt_)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
a, b = run(False), run(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): hx.cuda(non_blocking=True); ht.cuda(non_blocking=True)
torch.cuda.synchronize(); c = (time.perf_counter() - t0) / 50
print(f"resident inputs: {a*1e3:.2f} ms/step = {B/a:.0f} windows/s; pinned host inputs copied every step: {b*1e3:.2f} ms/step = {B/b:.0f} windows/s; "
      f"the two copies alone (2 x {B*L*4/1e6:.1f} MB): {c*1e3:.3f} ms = {2*B*L*4/c/1e9:.1f} GB/s")
