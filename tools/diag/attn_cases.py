"""Errors of the attention operator against fp64 for a list of shapes (the cases of tests/test_gpu_attention.py, printed
instead of asserted).   python tools/diag/attn_cases.py [N,H,Len,B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_attention as T
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (128, 8, 8, 5), (128, 8, 0, 3), (64, 16, 4, 5), (64, 16, 0, 5), (32, 32, 0, 5), (32, 8, 8, 5), (64, 2, 32, 3), (32, 2, 24, 3),
    (512, 2, 32, 3), (256, 4, 16, 5), (256, 4, 0, 3), (512, 2, 0, 2)]
for s in shapes:
    e = T._case(*s, seed=s[0] + s[2])
    print(s, " ".join(f"{k}={v:.2e}" for k, v in e.items()), flush=True)
