"""Relative L2 differences of the HIP path from the fp64 oracle (full variant, 4 x 2 x 512) with the wide levels on fp16-pair
products (default) and with every product on the fp32 MFMA (the switch f16_split = 0): python tools/diag/parity_numbers.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/oracle"); sys.path.insert(0, ROOT + "/tests")
import numpy as np
from parity_util import run_parity
for split in ("64", "0"):
    from ecg_denoise_amd import _lib
    _lib.apply_options("f16_split=" + split)
    res, _, _ = run_parity("full", 2, 512, 4)
    g = [v for k, v in res.items() if k.startswith("grad:")]
    a = [v for k, v in res.items() if k.startswith("act:")]
    print("f16_split", split, "y %.2e" % res["y"], "max act %.2e" % max(a), "grads: max %.2e median %.2e" % (max(g), float(np.median(g))))
