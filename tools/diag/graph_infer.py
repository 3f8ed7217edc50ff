"""Eval forward: eager launches vs hipGraph replay (ecg_denoise_amd.infer.GraphedForward), bench shape, same box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from ecg_denoise_amd import RALENet, _lib
from ecg_denoise_amd.infer import GraphedForward
_lib.apply_options(os.environ.get("RAL_TOOL_OPTIONS", ""))
B, L, n = 2048, 512, 100
m = RALENet("full", leads=1, L=L, max_batch=B, train=("GI_TRAIN" in os.environ), device="cuda:0", seed=1)
m.eval()
x = torch.randn(B, 1, L, device="cuda:0")
def timed(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    e = timed(lambda: m(x))
    gf = GraphedForward(m, B)
    g = timed(lambda: gf.graph.replay())
    print(f"eager {e:.3f} ms ({B / e:.0f} k windows/s)   graph replay {g:.3f} ms ({B / g:.0f} k)", flush=True)
