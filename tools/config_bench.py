"""Throughput of the BASELINE configurations that bench.py does not time (one MI355X each):
  C3  12-lead 1024-sample windows through newrale (inner RA-LENet 'full', 2 x 1024), batch 256 per GPU, train step
  C4  inference: 30-minute 2-lead records (650 000 samples) streamed in batches of 4096 512-sample windows,
      hipGraph-captured eval-mode forward, z-score / stitch included
Prints one JSON object per configuration."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ecg_denoise_amd import NewRALE, RALENet
from ecg_denoise_amd.infer import StreamingDenoiser
DEV = "cuda:0"
sync = torch.cuda.synchronize

# ---- C3 ----
B = 256
inner = RALENet("full", leads=2, L=1024, max_batch=B, device=DEV, seed=1)
m = NewRALE(inner, seed=2); m.train()
g = torch.Generator().manual_seed(2023)
x = torch.randn(B, 12, 1024, generator=g).to(DEV); t = torch.randn(B, 12, 1024, generator=g).to(DEV)
for _ in range(3): m.train_step(x, t)
sync(); t0 = time.perf_counter(); n = 20
for _ in range(n): m.train_step(x, t)
sync(); dt = (time.perf_counter() - t0) / n
print(json.dumps({"config": "C3 newrale 12-lead x 1024, batch 256, train step (adapter trains, inner RA-LENet frozen: forward + input-gradient backward)",
                  "ms_per_step": round(dt * 1e3, 3), "windows_per_s": round(B / dt, 1), "dtype": "f32"}))
del m, inner; torch.cuda.empty_cache()

# ---- C4 ----
model = RALENet("full", leads=2, L=512, max_batch=4096, train=False, device=DEV, seed=1)
sd = StreamingDenoiser(model, batch=4096, overlap=0, use_graph=True)
rng = np.random.default_rng(0)
recs = [torch.tensor((1000 + 100 * rng.standard_normal((2, 650000))).astype(np.float32), device=DEV) for _ in range(4)]
big = torch.cat(recs, dim=1)                      # 4 records back to back = 5078 windows: one full graph batch + remainder
for _ in range(2): sd.denoise(big)
sync(); t0 = time.perf_counter(); n = 10
for _ in range(n): y = sd.denoise(big)
sync(); dt = (time.perf_counter() - t0) / n
nwin = big.shape[1] // 512
print(json.dumps({"config": "C4 streaming inference, 4 x 30-min 2-lead records (650 000 samples each) per call, batch 4096 hipGraph forward, z-score + stitch included",
                  "ms_per_call": round(dt * 1e3, 2), "windows_per_s": round(nwin / dt, 1), "records_per_s": round(4 / dt, 2),
                  "realtime_factor": round(4 * 1800 / dt, 0), "dtype": "f32"}))
