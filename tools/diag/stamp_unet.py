"""Per-phase cycle shares of the fused U-Net inference kernel (diagnostic).  Build:
make -C ecg_denoise_amd/csrc STAMP=1 STAMPTU=UNET; run on the GPU box: python tools/diag/stamp_unet.py
Slot i = cycles thread 0 of workgroup 0 spent before RAL_STAMP_AT(i) (slot 1 = staging, slot l + 2 = layer l)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RAL_LIB_PATH"] = os.path.join(ROOT, "tools", "diag", "libralenet_stamp1.so")
sys.path.insert(0, ROOT)
import torch
from ecg_denoise_amd import UNet, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
m = UNet(leads=2, L=512, max_batch=B, train=False, device="cuda:0", seed=1)
m.eval()
x = torch.randn(B, 2, 512, device="cuda:0")
lib = _lib.lib()
fn = lib.ral_debug_stamps_unet
fn.argtypes = [C.c_void_p, C.c_int]
for _ in range(3): m(x)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
fn(buf, 1)
m(x); torch.cuda.synchronize()
fn(buf, 0)
tot = sum(buf[i] for i in range(32))
print("total cycles of workgroup 0:", tot)
for i in range(32):
    if buf[i]: print(f"  slot {i:2d} {buf[i]:12d}  {100.0*buf[i]/tot:5.1f}%")
