import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ecg_denoise_amd import UNet, _lib
B=2048
m = UNet(leads=2, L=512, max_batch=B, train=False, device="cuda:0", seed=1); m.eval()
_lib.check(_lib.lib().ral_set_option(m.eng.h, b"unet_fused", 0))
x = torch.randn(B, 2, 512, device="cuda:0")
for _ in range(5): m(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    m(x)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g):
    y = m(x)
for _ in range(5): g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): g.replay()
e1.record(); torch.cuda.synchronize()
dt = e0.elapsed_time(e1) / 200 * 1e-3
print(os.environ.get("RAL_UNET_EVAL_GRID"), "staged eval forward %.1f us, frac %.3f" % (dt * 1e6, B * 27 * 4096 / dt / 8e12))
