"""Round-6 additions to the golden fixtures (build container only; imports the REFERENCE from /root/reference through
oracle/gen_golden.py's recipe): whole-model cases at window lengths that are multiples of 16 but NOT of 256 - the lengths the
reference accepts (raletransformer.py:170,448-450: four PatchMerging halvings, positional table up to 1000) and the HIP path
refused through round 5.

  tests/golden/g3_nra_l2_L320.npz    raletransformer.ralenet (no R-wave bias: runs at any such length as it stands) at (3, 2, 320)
  tests/golden/g3_nra_l2_L128.npz    ... at (4, 2, 128): eight tokens at the bottleneck level
  tests/golden/g3_full_l2_L640.npz   transformer.ralenet(high_level_enhence=True) with SURVEY 8c's constructor-level patch
                                     (rwattn.whole_length = L >> i: the R-wave window centred in the level's tokens) at (2, 2, 640)

Same content as the round-1 G3 cases (gen_golden.model_case): train / eval outputs, loss, gradient summaries, BatchNorm running
statistics, three Adam steps.

    python oracle/gen_golden_r6.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ra, tr, un, newrale, dt, ev = G.import_reference()
    for (variant, leads, L, B) in (("nra", 2, 320, 3), ("nra", 2, 128, 4), ("full", 2, 640, 2)):
        gg = torch.Generator().manual_seed(2023)
        x = torch.randn(B, leads, L, generator=gg)
        tgt = torch.randn(B, leads, L, generator=gg)
        m, p = G.build_ralenet(ra, tr, variant, leads, L, seed=1234)
        G.model_case(f"g3_{variant}_l{leads}_L{L}", m, p, x, tgt, ["conv1.2"])


if __name__ == "__main__":
    main()
