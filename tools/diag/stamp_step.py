"""Per-phase cycle shares of the stamped kernels of one translation unit during one training step (diagnostic).
Build: make -C ecg_denoise_amd/csrc STAMP=<tag> STAMPTU=FWD|BWD|DW STAMPCOND='<expr on the template parameters, e.g. C==16>'
Run on the GPU box: STAMP_C=<tag> python tools/diag/stamp_step.py fwd|bwd|dw [fwdonly]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RAL_LIB_PATH"] = os.path.join(ROOT, "tools", "diag", "libralenet_stamp%s.so" % os.environ.get("STAMP_C", "1"))
os.environ["RAL_LANES"] = "1"; os.environ["RAL_NO_SIDE_STREAM"] = "1"
sys.path.insert(0, ROOT)
import torch
from ecg_denoise_amd import RALENet, _lib
B = 2048
m = RALENet("full", leads=1, L=512, max_batch=B, device="cuda:0", seed=1)
x = torch.randn(B, 1, 512, device="cuda:0")
lib = _lib.lib()
tu = sys.argv[1] if len(sys.argv) > 1 else "bwd"
fn = getattr(lib, {"fwd": "ral_debug_stamps_fwd", "bwd": "ral_debug_stamps", "dw": "ral_debug_stamps_dw"}[tu])
fn.argtypes = [C.c_void_p, C.c_int]
m.train()
def step():
    y = m(x); m.backward(torch.randn_like(y) / y.numel())
for _ in range(2): step()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
fn(buf, 1)
step(); torch.cuda.synchronize()
fn(buf, 0)
tot = sum(buf[i] for i in range(32))
print("total stamped cycles (workgroup 0 of every stamped launch of one step):", tot)
for i in range(32):
    if buf[i]: print(f"  slot {i:2d} {buf[i]:12d}  {100.0*buf[i]/tot:5.1f}%")
