import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import torch
import test_gpu_configs as T
for keys in (None, ("attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")):
    for scale in (1e-6, 1e-3, 1.0, 1e3):
        for f in (None, 0):
            kw = {} if keys is None else {"keys": keys}
            ey, eg, k, loss = T._scaled_parity(scale, f16_split=f, **kw)
            print(f"keys={'all5' if keys is None else 'proj/fc1/fc2'} scale={scale:g} f16_split={f}: y {ey:.2e} worst grad {eg:.2e} ({k}) loss {loss:.4f}", flush=True)
for s in (3e4, 1e6):
    for f in (None, 0):
        print(s, f, T._scaled_parity(s, keys=("transformer.blocks.0.mlp.fc1.weight",), f16_split=f), flush=True)
