"""Is the host ahead of the GPU?  Enqueue K training steps without synchronising: the wall time until the last call returns
(host enqueue) against the wall time until the GPU is done.  Also the host time of the single calls of one step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ecg_denoise_amd import RALENet
from ecg_denoise_amd.dp import DataParallelTrainer, HipEngineAdapter

dev = "cuda:0"
B, K = 2048, 20
m = RALENet("full", leads=1, L=512, max_batch=B, train=True, device=dev, seed=2023)
tr = DataParallelTrainer(HipEngineAdapter(m))
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, 1, 512, device=dev, generator=g); y = torch.randn(B, 1, 512, device=dev, generator=g)
for _ in range(5): tr.train_step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): tr.train_step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms per step, GPU done after %.2f ms per step" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
e = tr.e
torch.cuda.synchronize()
ts = [time.perf_counter()]
e.forward_begin(x); ts.append(time.perf_counter())
pred = e.forward_end(B); ts.append(time.perf_counter())
loss, snr, rmse = e.loss(pred, y, B); ts.append(time.perf_counter())
e.backward_begin(); ts.append(time.perf_counter())
e.backward_end(B); ts.append(time.perf_counter())
e.adam(1e-3); ts.append(time.perf_counter())
torch.cuda.synchronize()
names = ["forward_begin", "forward_end", "loss", "backward_begin", "backward_end", "adam"]
print("host ms per call on an idle GPU: " + ", ".join("%s %.2f" % (n, (b - a) * 1e3) for n, a, b in zip(names, ts, ts[1:])))
