import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
if len(sys.argv) > 1:
    import torch
    from ecg_denoise_amd import RALENet
    m = RALENet("full", leads=2, L=512, max_batch=2, train=True, device="cuda:0", seed=3)
    x = torch.randn(2, 2, 512, generator=torch.Generator().manual_seed(1)).cuda()
    m.train(); y = m(x)
    out = {k: m.debug_tensor(k).cpu() for k in ("blk0.out", "blk2.out", "blk4.out")}
    torch.save(out, sys.argv[1])
    sys.exit(0)
import torch
for v in ("0", "1"):
    subprocess.run([sys.executable, __file__, f"/tmp/mlpw_{v}.pt"], env=dict(os.environ, RAL_MLP_FWD_W=v), check=True)
a, b = torch.load("/tmp/mlpw_0.pt"), torch.load("/tmp/mlpw_1.pt")
for k, (N, C) in {"blk0.out": (512, 8), "blk2.out": (256, 16), "blk4.out": (128, 32)}.items():
    d = (a[k] - b[k]).reshape(2, N, C).abs()
    print(k, "max", d.max().item(), "rel", (d.norm() / a[k].norm()).item())
    bad = (d[0] > 1e-5).nonzero()
    toks = sorted(set(bad[:, 0].tolist())); chans = sorted(set(bad[:, 1].tolist()))
    print("   bad tokens", toks[:40], "... n =", len(toks), " channels", chans)
