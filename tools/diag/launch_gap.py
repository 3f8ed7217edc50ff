"""Back-to-back dependent kernels in one stream: time per launch of a trivial kernel (eager and as a hipGraph replay), i.e.
the floor under the inter-kernel gaps of the lane chains (tools/diag/step_timeline.py)."""
import torch
x = torch.zeros(64, device="cuda")
def run(n):
    for _ in range(n):
        x.add_(1.0)
for _ in range(3): run(100)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(2000); e1.record(); torch.cuda.synchronize()
print("eager: %.2f us per launch" % (e0.elapsed_time(e1) * 1e3 / 2000))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(10)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        run(2000)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print("hipGraph replay: %.2f us per kernel node" % (e0.elapsed_time(e1) * 1e3 / 2000))
