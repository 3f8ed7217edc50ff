"""Timing of the GPU windowing / noise-mixing step (ral_prep_windows) next to the oracle on the host cores
(lives under tests/ because it runs the oracle; not collected by pytest).  python tests/bench_dataprep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from ecg_denoise_amd.data import prep_windows
import dataprep_oracle as D
rng = np.random.default_rng(0)
for T in (256 * 32, 650000 // 256 * 256, 256 * 65536):
    sig = (1000 + 100 * rng.standard_normal((T, 2))).astype(np.float32); noise = (20 * rng.standard_normal((T, 2))).astype(np.float32)
    s, n = torch.tensor(sig, device="cuda:0"), torch.tensor(noise, device="cuda:0")
    for _ in range(3): prep_windows(s, n, 0.0, 256)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 50
    for _ in range(reps): prep_windows(s, n, 0.0, 256)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    t1 = time.perf_counter(); D.prep_segment(sig, noise, 0.0, 256); dc = time.perf_counter() - t1
    bytes_alg = T * 2 * 4 * (2 + 2 + 2)   # two read passes of both inputs, two outputs
    print(f"T={T:9d} ({T//256} windows): GPU {dt*1e6:8.1f} us = {T/256/dt/1e6:7.2f} M windows/s, {bytes_alg/dt/1e9:7.1f} GB/s algorithmic; "
          f"oracle on the host {dc*1e3:8.2f} ms")
