#!/bin/bash
# Environment-knob sweep on the GPU box: each line is one bench.py run (20 steps) with the given variables.
#   gpurun -- 'bash tools/sweep.sh'
run() { echo "=== $*"; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu --kinds 2>&1 | grep -E "^  (dw|mlp_bwd)|value" | cut -c1-110; }
run RAL_X=default
run RAL_FUSE_DW=0
run RAL_FUSE_DW=16
run RAL_LANES=1
run RAL_LANES=1 RAL_NO_SIDE_STREAM=1
run RAL_DW_KSPLIT=512,512,256,128,64
