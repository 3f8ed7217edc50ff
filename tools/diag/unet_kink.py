"""Is the 1e-3 .. 3e-3 gradient deviation of the U-Net at large batch the LeakyReLU kink (fp32 rounding flips a slope), i.e.
does an fp32 CPU evaluation of the oracle deviate from the fp64 one by as much as the HIP path does?"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "oracle")
from collections import OrderedDict
import torch, numpy as np
import ralenet_oracle as O
from parity_util import rel
import test_gpu_unet as T
L, B = int(sys.argv[1]), int(sys.argv[2])
m, y, loss, p, bn, yo, lo, grads, x, tgt = T._run(2, L, B, seed=4321)
p32 = OrderedDict((k, v.detach().float().requires_grad_(True)) for k, v in p.items())
bn32 = O.unet_bn_state(p32, torch.float32)
y32 = O.unet_forward(p32, x.float(), True, bn32)
g32 = torch.autograd.grad(O.mse(y32, tgt.float()), list(p32.values()))
ng = m.named_grads()
rows = []
for (k, _), g64, gf in zip(p.items(), grads, g32):
    if g64.norm().item() < 1e-9: continue
    rows.append((rel(ng[k].cpu().numpy(), g64.numpy()), rel(gf.numpy(), g64.numpy()), k))
rows.sort(reverse=True)
print("L", L, "B", B, " worst keys:  HIP vs fp64 | CPU-fp32 vs fp64")
for a, b, k in rows[:8]: print("  %.2e  %.2e  %s" % (a, b, k))
