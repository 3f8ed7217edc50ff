#!/bin/bash
# Environment-knob sweep on the GPU box: each line is one bench.py run (20 steps) with the given variables.
#   gpurun -- 'bash tools/sweep.sh'
run() { echo "=== $*"; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu --no-infer --kinds 2>&1 | grep -E "^  (dw|mlp_bwd)|value" | cut -c1-110; }
run RAL_X=default
run RAL_DW_KSPLIT=256,256,256,256,128
run RAL_DW_KSPLIT=256,256,256,256,256
run RAL_DW_KSPLIT=256,256,256,192,96
run RAL_DW_KSPLIT=512,512,512,256,128
run RAL_DW_KSPLIT=256,256,256,256,128 RAL_DW_LDS=49152
run RAL_DW_KSPLIT=256,256,256,384,192 RAL_DW_LDS=49152
