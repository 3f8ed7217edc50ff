"""Per-phase cycles of one instantiation of the U-Net backward stage kernel (diagnostic).  Build, e.g.:
  make -C ecg_denoise_amd/csrc STAMP=1 STAMPTU=UNET STAMPSEL='CIN==4&&COUT==2&&MODE==2'
run on the GPU box: python tools/diag/stamp_unet_bwd.py.  Slots (thread 0 of workgroup 0, summed over its passes):
16 prologue, 17 load phase, 18 input gradient, 19 weight gradient, 20 barrier wait, 21 flush."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RAL_LIB_PATH"] = os.path.join(ROOT, "tools", "diag", "libralenet_stamp%s.so" % os.environ.get("STAMP", "1"))
sys.path.insert(0, ROOT)
import torch
from ecg_denoise_amd import UNet, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
m = UNet(leads=2, L=512, max_batch=B, train=True, device="cuda:0", seed=1)
m.train()
x = torch.randn(B, 2, 512, device="cuda:0"); t = torch.randn(B, 2, 512, device="cuda:0")
lib = _lib.lib()
fn = lib.ral_debug_stamps_unet
fn.argtypes = [C.c_void_p, C.c_int]
for _ in range(3): m.train_step(x, t)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
fn(buf, 1)
m.train_step(x, t); torch.cuda.synchronize()
fn(buf, 0)
names = {16: "prologue", 17: "load phase", 18: "input gradient", 19: "weight gradient", 20: "barrier", 21: "flush"}
tot = sum(buf[i] for i in range(16, 22))
print("backward stage, cycles of the stamped workgroup:", tot)
for i in range(16, 22):
    print(f"  {names[i]:16s} {buf[i]:10d}  {100.0 * buf[i] / max(tot, 1):5.1f}%")
names = {8: "prologue", 9: "load phase", 10: "compute", 11: "barrier", 12: "flush"}
tot = sum(buf[i] for i in range(8, 13))
print("forward stage, cycles of the stamped workgroup:", tot)
for i in range(8, 13):
    print(f"  {names[i]:16s} {buf[i]:10d}  {100.0 * buf[i] / max(tot, 1):5.1f}%")
