"""Round-2 additions to the golden fixtures (build container only; imports the REFERENCE from /root/reference through
oracle/gen_golden.py's recipe).  Writes, next to the round-1 fixtures:

  tests/golden/g3_acdae_l2_L512.npz   the ACDAE comparison baseline (model/ACDAE.py:62-86) at (4, 2, 512): outputs, loss,
                                      gradient summaries, three Adam steps (gen_golden.model_case)
  tests/golden/g3_newrale_L1024.npz   BASELINE config 4 at its stated window: `newrale` (ralenet_12leads.py:680-709)
                                      around the reference `transformer.ralenet(high_level_enhence=True)` patched to
                                      L = 1024 (PE max_len, rwattn.whole_length: SURVEY 8c), input (2, 12, 1024):
                                      train-mode output, loss, the eight adapter gradients, the inner BatchNorm's
                                      running statistics after the step (quirk A16), eval-mode output.

    python oracle/gen_golden_r2.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
import ralenet_oracle as O  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(3)
    ra, tr, un, newrale, dt, ev = G.import_reference()
    L = 1024
    inner, _ = G.build_ralenet(ra, tr, "full", 2, L, seed=1234)
    nr = newrale(inner)
    pa = O.init_params(O.newrale_param_shapes(), 77)
    nr.load_state_dict(pa, strict=False)
    gg = torch.Generator().manual_seed(2023)
    x = torch.randn(2, 12, L, generator=gg); tgt = torch.randn(2, 12, L, generator=gg)
    nr.train()
    y = nr(x); loss = torch.nn.functional.mse_loss(y, tgt); loss.backward()
    nm = dict(nr.named_parameters())
    out = {"x": x.numpy(), "target": tgt.numpy(), "y_train": y.detach().numpy(), "loss": np.float64(loss.item()),
           "snr": ev.SNR(tgt, y.detach()).numpy(), "rmse": ev.RMSE(tgt, y.detach()).numpy()}
    for k in pa:
        out["grad_" + k] = nm[k].grad.numpy()
    sd = inner.state_dict()
    out["bn_mean"] = sd["conv1.2.running_mean"].numpy().copy()
    out["bn_var"] = sd["conv1.2.running_var"].numpy().copy()
    assert all(p.grad is None for k, p in inner.named_parameters()), "inner model must stay frozen"
    nr.eval()
    with torch.no_grad():
        out["y_eval"] = nr(x).numpy()
    np.savez_compressed(os.path.join(G.OUT, "g3_newrale_L1024.npz"), **out)
    print("g3_newrale_L1024: loss", loss.item(), "snr", out["snr"], "rmse", out["rmse"])

    # ---- ACDAE (model/ACDAE.py), the comparison baseline of SURVEY 8f-4: (4, 2, 512), same recipe as the G3 cases ----
    import importlib
    ac = importlib.import_module("model.ACDAE")
    m = ac.ACDAE()
    p = O.init_params(O.acdae_param_shapes(), 1234)
    assert list(p.keys()) == [k for k, _ in m.named_parameters()], "state_dict order"
    m.load_state_dict(p)
    gg = torch.Generator().manual_seed(2023)
    x = torch.randn(4, 2, 512, generator=gg); tgt = torch.randn(4, 2, 512, generator=gg)
    G.model_case("g3_acdae_l2_L512", m, p, x, tgt, [])


if __name__ == "__main__":
    main()
