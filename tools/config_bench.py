import os
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
for use_graph in (True, False):
    sd = StreamingDenoiser(model, batch=4096, overlap=0, use_graph=use_graph)
    rng = np.random.default_rng(0)
    R = 8                                         # 8 records = 10 160 windows: two full batches of 4096 + a remainder
    group = torch.tensor((1000 + 100 * rng.standard_normal((R, 2, 650000))).astype(np.float32), device=DEV)
    for _ in range(2): sd.denoise(group)
    sync(); t0 = time.perf_counter(); n = 10
    for _ in range(n): y = sd.denoise(group)
    sync(); dt = (time.perf_counter() - t0) / n
    nwin = R * sd.windows_per_record(650000)
    print(json.dumps({"config": f"C4 streaming inference, {R} x 30-min 2-lead records (650 000 samples each) per call, batches of 4096 windows, "
                                f"HIP window/z-score and de-normalise/stitch kernels, {'one hipGraph per record group' if use_graph else 'eager launches'}",
                      "ms_per_call": round(dt * 1e3, 2), "windows_per_s": round(nwin / dt, 1), "records_per_s": round(R / dt, 2),
                      "realtime_factor": round(R * 1800 / dt, 0), "dtype": "f32"}))
