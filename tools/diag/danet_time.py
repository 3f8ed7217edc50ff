import torch, time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from ecg_denoise_amd import DANet
B=2048
m=DANet(L=512,max_batch=B,device="cuda:0",seed=1)
x=torch.randn(B,2,512,device="cuda:0"); t=torch.randn_like(x)
def step():
    y=m(x); m.loss_and_metrics(y,t); m.backward(); m.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize(); print("train ms", round((time.perf_counter()-t0)/30*1e3,3))
