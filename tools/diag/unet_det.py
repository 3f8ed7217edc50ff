"""Run-to-run spread of the U-Net gradients of one backward pass and of the parameters after three Adam steps (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from ecg_denoise_amd import UNet
Bg, L = 128, 256
g = torch.Generator().manual_seed(77)
x = torch.randn(Bg, 2, L, generator=g).cuda(); t = torch.randn(Bg, 2, L, generator=g).cuda()
grads, states = [], []
for rep in range(6):
    m = UNet(leads=2, L=L, max_batch=Bg, device="cuda:0", seed=100)
    m.train()
    y = m(x); m.loss_and_metrics(y, t); m.backward()
    grads.append({k: v.cpu().clone() for k, v in m.named_grads().items()})
    m.step()
    for _ in range(2): m.train_step(x, t)
    states.append({k: v.cpu().clone() for k, v in m.state_dict().items()})
rel = lambda a, b: ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
for rep in range(1, 6):
    wg = max((rel(grads[rep][k], grads[0][k]), k) for k in grads[0] if not k.endswith("conv.bias"))
    ws = max((rel(states[rep][k], states[0][k]), k) for k in states[0] if states[0][k].dtype.is_floating_point and not k.endswith("conv.bias") and not k.endswith("running_mean"))
    print("rep", rep, "grad worst", "%.2e" % wg[0], wg[1], "| state worst", "%.2e" % ws[0], ws[1], "| g(bn0.bias)", grads[rep]["EncList.0.bn.bias"].tolist())
