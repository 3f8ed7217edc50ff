import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import test_gpu_configs as T
for rep in range(3):
    for f16 in (None, 0):
        ey, eg, k, loss = T._scaled_parity(1e-3, **({} if f16 is None else {"f16_split": f16}))
        print(rep, "f16_split", f16, "ey %.2e worst %.2e %s own %.2e loss %.6f" % (ey, eg, k, T._scaled_parity.own, loss), flush=True)
