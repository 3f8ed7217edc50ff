import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ecg_denoise_amd import DANet
B = 2048
m = DANet(L=512, max_batch=B, device="cuda:0", seed=1); m.train()
x = torch.randn(B, 2, 512, device="cuda:0"); t = torch.randn(B, 2, 512, device="cuda:0")
for _ in range(5): m.train_step(x, t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): m.train_step(x, t)
torch.cuda.synchronize()
print(os.environ.get("RAL_DANET_GRID_D"), "danet train %.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3))
