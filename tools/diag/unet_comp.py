"""Component-wise error of the U-Net gradients against the fp64 oracle (diagnostic): for every parameter tensor the largest
absolute component error relative to the tensor's largest component, over several repeats."""
import os, sys, torch
from collections import OrderedDict
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ralenet_oracle as O
from ecg_denoise_amd import UNet
B, L = 128, 256
p32 = O.init_params(O.unet_param_shapes(2), 100)
g = torch.Generator().manual_seed(77)
x = torch.randn(B, 2, L, generator=g); t = torch.randn(B, 2, L, generator=g)
p = OrderedDict((k, v.double().requires_grad_(True)) for k, v in p32.items())
bn = O.unet_bn_state(p, torch.float64)
yo = O.unet_forward(p, x.double(), True, bn)
lo = O.mse(yo, t.double())
go = dict(zip(p.keys(), torch.autograd.grad(lo, list(p.values()))))
for rep in range(4):
    m = UNet(leads=2, L=L, max_batch=B, device="cuda:0"); m.load_state_dict(p32, strict=False); m.train()
    y = m(x.cuda()); m.loss_and_metrics(y, t.cuda()); m.backward(); torch.cuda.synchronize()
    worst = []
    for k, v in m.named_grads().items():
        if k.endswith("conv.bias"): continue
        e = (v.cpu().double() - go[k]).abs().max().item() / go[k].abs().max().item()
        worst.append((e, k))
    worst.sort(reverse=True)
    print("rep", rep, [(f"{e:.1e}", k) for e, k in worst[:4]])
