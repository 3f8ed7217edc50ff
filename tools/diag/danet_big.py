"""Per-window input-gradient error of the DANet backward against the fp64 oracle (diagnostic): isolates the windows in
which a non-differentiable point (ReLU at 0, a max-pool tie) is resolved differently in fp32 and fp64."""
import os, sys, torch
from collections import OrderedDict
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import danet_oracle as D
from ecg_denoise_amd import DANet
B, L, SEED = 256, 512, 7
st64 = D.init_state(SEED, dtype=torch.float64)
st32 = OrderedDict((k, v.clone().float() if v.dtype.is_floating_point else v.clone()) for k, v in st64.items())
gg = torch.Generator().manual_seed(11)
x = torch.randn(B, 2, L, generator=gg); tgt = torch.randn(B, 2, L, generator=gg)
xd = x.double().requires_grad_(True)
params = OrderedDict((k, v.requires_grad_(True)) for k, v in st64.items() if D.is_param(k) and ".dam.fcn2." not in k)
y64 = D.danet_forward(st64, xd, training=True)
torch.nn.functional.mse_loss(y64, tgt.double()).backward()
m = DANet(L=L, max_batch=B, device="cuda:0"); m.load_state_dict(st32); m.train()
y = m(x.cuda()); m.loss_and_metrics(y, tgt.cuda()); dx = m.backward(want_dx=True).cpu().double()
e = (dx - xd.grad).flatten(1).norm(dim=1) / xd.grad.flatten(1).norm(dim=1)
top = torch.topk(e, 6)
print("per-window dx rel err: median %.1e" % e.median().item(), "top:", [(int(i), "%.1e" % v) for v, i in zip(top.values, top.indices)])
ey = (y.cpu().double() - y64.detach()).flatten(1).norm(dim=1) / y64.detach().flatten(1).norm(dim=1)
print("per-window y rel err max %.1e at %d" % (ey.max().item(), int(ey.argmax())))
