import os, sys, time
sys.path.insert(0, "oracle")
from collections import OrderedDict
import torch, ralenet_oracle as O
print("cpu_count", os.cpu_count())
for B in (32,):
  for th in (1, 2, 4, 8, 16, 32, 64):
    torch.set_num_threads(th)
    p = O.init_params(O.ralenet_param_shapes("full", 1), 1)
    g = torch.Generator().manual_seed(2023)
    x = torch.randn(B, 1, 512, generator=g); tgt = torch.randn(B, 1, 512, generator=g)
    m = OrderedDict((k, torch.zeros_like(v)) for k, v in p.items()); v = OrderedDict((k, torch.zeros_like(t)) for k, t in p.items())
    bn = O.new_bn_state()
    fwd = lambda pp, xx: O.ralenet_forward(pp, xx, "full", True, bn)
    O.train_step(p, x, tgt, fwd, m, v, 1)
    n, t0 = 0, time.time()
    while time.time() - t0 < 4.0:
        O.train_step(p, x, tgt, fwd, m, v, n + 2); n += 1
    print("B", B, "threads", th, round(B * n / (time.time() - t0), 1), "windows/s", flush=True)
