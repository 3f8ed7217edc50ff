import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
import ralenet_oracle as O
from ecg_denoise_amd import RALENet
B, L = 2, 256
m = RALENet("nra", leads=2, L=L, max_batch=B, device="cuda:0")
m.load_state_dict(O.init_params(O.ralenet_param_shapes("nra", 2), 1234), strict=False)
x = torch.randn(B, 2, L, device="cuda:0")
m.train(); y = m(x); torch.cuda.synchronize()
N, C, H = L, 8, 2
qkv = m.debug_tensor("blk0.qkv")[:B*3*N*C].view(B, 3, H, N, 4).cpu().double()
o = m.debug_tensor("blk0.o")[:B*N*C].view(B, H, N, 4).cpu().double()
q, k, v = qkv[:,0], qkv[:,1], qkv[:,2]
ref = torch.softmax(q @ k.transpose(-1,-2), -1) @ v
print("rel err", ((o-ref).norm()/ref.norm()).item())
d = (o-ref).abs()
print("max err per d:", d.amax((0,1,2)))
print("err by query idx (first 20):", d.amax((0,1,3))[:20])
print(o[0,0,:4]); print(ref[0,0,:4])
