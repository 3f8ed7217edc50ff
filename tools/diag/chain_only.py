"""Training step with and without the weight-gradient kernels (ral_backward vs ral_backward_input): how much of the
side-stream work the schedule fails to hide (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from ecg_denoise_amd import RALENet
B = 2048
m = RALENet("full", leads=1, L=512, max_batch=B, device="cuda:0", seed=1)
x = torch.randn(B, 1, 512, device="cuda:0"); t = torch.randn_like(x)
def step(full):
    y = m(x); m.loss_and_metrics(y, t)
    if full: m.backward()
    else: m.backward_input(m._dy)
    m.step()
for full in (True, False, True, False):
    for _ in range(5): step(full)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): step(full)
    torch.cuda.synchronize()
    print("with dW" if full else "without dW", round((time.perf_counter() - t0) / 40 * 1e3, 3), "ms")
