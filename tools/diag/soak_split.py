"""Soak of the split-operand kernels: 300 training steps on fresh random batches (batch 256 x 2 x 512), the loss trajectory of
the default arithmetic next to the fp32-MFMA one (the switch f16_split = 0) from the same weights and data: finite everywhere, and the
two stay within the run-to-run noise of the fp32 atomics for the first steps.   python tools/diag/soak_split.py"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    from ecg_denoise_amd import _lib
    _lib.apply_options(os.environ.get("RAL_TOOL_OPTIONS", ""))
    import torch
    from ecg_denoise_amd import RALENet
    torch.manual_seed(0)
    m = RALENet("full", leads=2, L=512, max_batch=256, device="cuda:0", seed=3)
    m.train()
    g = torch.Generator(device="cuda:0").manual_seed(1)
    out = []
    for i in range(300):
        x = torch.randn(256, 2, 512, device="cuda:0", generator=g)
        t = 0.5 * x + 0.1 * torch.randn(256, 2, 512, device="cuda:0", generator=g)
        out.append(float(m.train_step(x, t)["loss"]))
    print(json.dumps(out))
else:
    res = {}
    for split in ("64", "0"):
        env = dict(os.environ, RAL_TOOL_OPTIONS="f16_split=" + split)
        r = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True)
        if r.returncode != 0 or not r.stdout.strip():
            sys.exit("run failed:\n" + r.stderr[-2000:])
        res[split] = json.loads(r.stdout.strip().splitlines()[-1])
    a, b = res["64"], res["0"]
    import math
    assert all(math.isfinite(v) for v in a + b)
    for i in (0, 1, 2, 5, 10, 20, 50, 100, 200, 299):
        print(f"step {i:3d}: split {a[i]:.6f}   fp32 MFMA {b[i]:.6f}   rel diff {abs(a[i] - b[i]) / abs(b[i]):.2e}")
