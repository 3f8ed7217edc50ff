#!/bin/bash
# Environment-knob sweep on the GPU box: each line is one bench.py run (30 steps) with the given variables.
#   gpurun -- 'bash tools/sweep.sh'
run() { echo "=== $*"; env "$@" python bench.py --steps 30 --warmup 3 --no-cpu --no-infer 2>&1 | grep -E "value" | sed -e 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/windows\/s \1  ms \2/'; }
run RAL_X=default
run RAL_DW_LDS=65536
run RAL_DW_LDS=49152
run RAL_DW_LDS=40960
run RAL_LANES=3
run RAL_LANES=1
run RAL_NO_SIDE_STREAM=1
run RAL_MLP_LDS=65536
run RAL_ATTN_BWD_LDS=65536
run RAL_FUSE_DW=16
run RAL_FUSE_DW=8
run RAL_X=default
