"""Print the HIP-vs-oracle error table for one configuration (run on the GPU box)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from parity_util import run_parity  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="full")
ap.add_argument("--leads", type=int, default=2)
ap.add_argument("--L", type=int, default=512)
ap.add_argument("--B", type=int, default=4)
a = ap.parse_args()
t0 = time.time()
res, model, _ = run_parity(a.variant, a.leads, a.L, a.B)
print(f"== {a.variant} leads={a.leads} L={a.L} B={a.B}  ({time.time()-t0:.1f}s)")
acts = [(k, v) for k, v in res.items() if not k.startswith("grad")]
for k, v in acts:
    print(f"  {k:28s} {v:.3e}")
grads = sorted([(v, k) for k, v in res.items() if k.startswith("grad")], reverse=True)
print("  worst gradients:")
for v, k in grads[:25]:
    print(f"  {k:70s} {v:.3e}")
print("  median grad err: %.3e" % sorted(v for v, _ in grads)[len(grads) // 2])
