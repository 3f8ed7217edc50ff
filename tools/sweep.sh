#!/bin/bash
# Environment-knob sweep on the GPU box: each line is one bench.py run (30 steps) with the given variables.
#   gpurun -- 'bash tools/sweep.sh'
run() { echo "=== $*"; env "$@" python bench.py --steps 30 --warmup 3 --no-cpu --no-infer 2>&1 | grep -E "value" | sed -e 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/windows\/s \1  ms \2/'; }
run RAL_X=default
run RAL_LANES=3
run RAL_LANES=4
run RAL_DW_KSPLIT=64,64,64,32,16
run RAL_ATTN_SPLIT=1
run RAL_MLP_LDS=100000
run RAL_FUSE_DW=16
run RAL_X=default
