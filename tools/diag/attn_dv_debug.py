import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import torch
from ecg_denoise_amd import _lib
import test_gpu_attention as T
N, H, Len, B = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 2, 0, 1)))
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B, 3 * H, N, 4, generator=g); qkv[:, :H] *= 0.5
do = torch.randn(B, H, N, 4, generator=g)
q, k, v = (t.double() for t in (qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]))
s = q @ k.transpose(-1, -2)
P = torch.softmax(s, -1)
dv_ref = P.transpose(-1, -2) @ do.double()
qd, dod = qkv.cuda(), do.cuda()
o = torch.empty(B, H, N, 4, device="cuda"); lse = torch.empty(B, H, N, device="cuda"); dqkv = torch.full_like(qd, float("nan"))
L = _lib.lib()
_lib.check(L.ral_attention_forward(T._vp(qd), T._vp(o), T._vp(lse), T._vp(None), N, H, 0, B, T._stream()))
ns = L.ral_attention_backward_scratch_floats(N, H, 0, 0, B)
sc = torch.empty(max(ns, 1), device="cuda")
_lib.check(L.ral_attention_backward(T._vp(qd), T._vp(o), T._vp(dod), T._vp(lse), T._vp(None), T._vp(None), T._vp(dqkv), T._vp(sc), ns, N, H, 0, B, T._stream()))
torch.cuda.synchronize()
dv = dqkv[:, 2 * H:].double().cpu()
for b in range(B):
    for h in range(H):
        e = (dv[b, h] - dv_ref[b, h])
        print("win", b, "head", h, "rel err per key tile:", [f"{(e[t:t+16].norm() / dv_ref[b, h, t:t+16].norm()).item():.1e}" for t in range(0, N, 16)])
        print("   per dim:", [f"{(e[:, d].norm() / dv_ref[b, h, :, d].norm()).item():.1e}" for d in range(4)])
print("ratio sample (got/ref) head 0 keys 0..3:\n", (dv[0, 0, :4] / dv_ref[0, 0, :4]))
# which P would explain dv?  solve dv = P^T dO for the contribution pattern: compare against dv computed with fp16-rounded P / fp16-rounded dO
for name, Pm, Dm in (("P->f16", P.half().double(), do.double()), ("dO->f16", P, do.half().double()), ("dO h1 only, P full", P, do.half().double())):
    alt = Pm.transpose(-1, -2) @ Dm
    print(name, "rel diff of that hypothesis from kernel:", ((dv - alt).norm() / alt.norm()).item())
# explain the error of each key tile as a combination of per-query-tile contributions (a = -1: missing, +1: doubled)
import numpy as np
b, h = 0, 0
for kt in range(N // 16):
    E = (dv[b, h, 16 * kt:16 * kt + 16] - dv_ref[b, h, 16 * kt:16 * kt + 16]).reshape(-1).numpy()
    Cs = []
    for qt in range(N // 16):
        Cq = P[b, h, 16 * qt:16 * qt + 16, 16 * kt:16 * kt + 16].transpose(-1, -2) @ do[b, h, 16 * qt:16 * qt + 16].double()
        Cs.append(Cq.reshape(-1).numpy())
    A = np.stack(Cs, 1)
    a, res, *_ = np.linalg.lstsq(A, E, rcond=None)
    print("key tile", kt, "coefficients per query tile:", np.round(a, 3), "residual", np.linalg.norm(A @ a - E) / (np.linalg.norm(E) + 1e-30))
# and as contributions of OTHER key tiles' P (wrong column block)
for kt in range(N // 16):
    E = (dv[b, h, 16 * kt:16 * kt + 16] - dv_ref[b, h, 16 * kt:16 * kt + 16]).reshape(-1).numpy()
    Cs = []
    for k2 in range(N // 16):
        Cq = P[b, h, :, 16 * k2:16 * k2 + 16].transpose(-1, -2) @ do[b, h].double()
        Cs.append(Cq.reshape(-1).numpy())
    A = np.stack(Cs, 1)
    a, res, *_ = np.linalg.lstsq(A, E, rcond=None)
    print("key tile", kt, "as other key tiles' dV:", np.round(a, 3), "residual", np.linalg.norm(A @ a - E) / (np.linalg.norm(E) + 1e-30))
torch.set_printoptions(precision=4, linewidth=200, sci_mode=False)
print("dv got - ref, head 0, last key tile (rows = keys, cols = dims):\n", (dv[0, 0, N - 16:] - dv_ref[0, 0, N - 16:]))
print("ref:\n", dv_ref[0, 0, N - 16:])
# per query-tile candidates for the LAST key tile: P_h1-only, P_h2-only contributions
kt = N // 16 - 1
E = (dv[0, 0, 16 * kt:] - dv_ref[0, 0, 16 * kt:])
for qt in range(N // 16):
    Pq = P[0, 0, 16 * qt:16 * qt + 16, 16 * kt:]            # (16 q, 16 k)
    dOq = do[0, 0, 16 * qt:16 * qt + 16].double()
    Kq = k[0, 0, 16 * qt:16 * qt + 16]; Vq = v[0, 0, 16 * qt:16 * qt + 16]; Qq = q[0, 0, 16 * qt:16 * qt + 16]
    for nm, X in (("dO", dOq), ("k", Kq), ("v", Vq), ("q", Qq)):
        C = Pq.transpose(-1, -2) @ X
        a = (C * E).sum() / (C * C).sum()
        print(f"  query tile {qt}: projection of the error on P^T {nm}: coefficient {a.item():+.4f}, explains {((a * C).norm() / E.norm()).item():.3f} of its norm")
