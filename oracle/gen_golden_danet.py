"""Golden fixture of the DANet baseline (build container only; imports the REFERENCE's model/DAM.py from /root/reference).

  tests/golden/g3_danet_L512.npz   Seq2Seq2 (DAM.py:341-349) at (6, 2, 512) with the build-owned state of
                                   oracle/danet_oracle.init_state(4321): train-mode output, MSE loss, every parameter
                                   gradient (summaries for the large ones), running statistics after the step, eval-mode
                                   output from the INITIAL state.

    python oracle/gen_golden_danet.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
import danet_oracle as D  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(2)
    from model.DAM import Seq2Seq2
    m = Seq2Seq2()
    B, L = 6, 512
    gg = torch.Generator().manual_seed(2023)
    x = torch.randn(B, 2, L, generator=gg); tgt = torch.randn(B, 2, L, generator=gg)
    m.train()
    with torch.no_grad():
        m(x)                                    # materialise the Lazy modules
    st = D.init_state(4321)
    assert list(m.state_dict().keys()) == list(st.keys()), "state_dict order"
    m.load_state_dict(st)
    m.eval()
    with torch.no_grad():
        y_eval = m(x).numpy().copy()
    m.train()
    y = m(x)
    loss = torch.nn.functional.mse_loss(y, tgt)
    loss.backward()
    out = {"x": x.numpy(), "target": tgt.numpy(), "y_eval": y_eval, "y_train": y.detach().numpy(), "loss": np.float64(loss.item())}
    for k, p in m.named_parameters():
        out["grad_" + k] = p.grad.numpy()
    sd = m.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
            out["after_" + k] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g3_danet_L512.npz"), **out)
    print("g3_danet_L512: loss", loss.item(), "keys", len(out), "params", sum(p.numel() for p in m.parameters()))


if __name__ == "__main__":
    main()
