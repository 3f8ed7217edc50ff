import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RAL_LIB_PATH"] = os.path.join(ROOT, "tools", "diag", "libralenet_stamp%s.so" % os.environ.get("STAMP_C", ""))
os.environ["RAL_LANES"] = "1"; os.environ["RAL_NO_SIDE_STREAM"] = "1"
sys.path.insert(0, ROOT)
import torch
from ecg_denoise_amd import RALENet, _lib
B = 2048
m = RALENet("full", leads=1, L=512, max_batch=B, device="cuda:0", seed=1)
x = torch.randn(B, 1, 512, device="cuda:0")
lib = _lib.lib()
lib.ral_debug_stamps.argtypes = [C.c_void_p, C.c_int]
m.train()
for _ in range(2): m(x)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib.ral_debug_stamps(buf, 1)
m(x); torch.cuda.synchronize()
lib.ral_debug_stamps(buf, 0)
names = ["load x,o", "proj GEMM", "x1 out + LN2", "fc1 GEMM", "u_pre out + A0", "GELU pass", "fc2 GEMM", "x2 out"]
tot = sum(buf[i] for i in range(8))
print("block 0 of every k_mlp_fwd launch of one forward (18 launches, all levels), cycles:")
for i, n in enumerate(names): print(f"  {n:16s} {buf[i]:10d}  {100.0*buf[i]/tot:5.1f}%")
