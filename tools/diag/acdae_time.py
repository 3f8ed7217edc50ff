"""ACDAE train-step timing loop for rocprofv3 (tools/diag/prof_cmd.sh tools/diag/acdae_time.py)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from ecg_denoise_amd import ACDAE
B = 2048
m = ACDAE(L=512, max_batch=B, device="cuda:0", seed=1)
x = torch.randn(B, 2, 512, device="cuda:0"); t = torch.randn_like(x)
for _ in range(3): m.train_step(x, t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): m.train_step(x, t)
torch.cuda.synchronize(); print("train ms", round((time.perf_counter() - t0) / 20 * 1e3, 3))
