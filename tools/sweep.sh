#!/bin/bash
# Environment-knob sweep on the GPU box: each line is one bench.py run (30 steps) with the given variables.
#   gpurun -- 'bash tools/sweep.sh'
run() { echo "=== $*"; env "$@" python bench.py --steps 30 --warmup 3 --no-cpu --no-infer 2>&1 | grep -E "value" | sed -e 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/windows\/s \1  ms \2/'; }
run RAL_X=default
run RAL_ATTN_BWD_LDS=65536
run RAL_ATTN_BWD_LDS=98304
run RAL_ATTN_FWD_LDS=49152
run RAL_ATTN_FWD_LDS=98304
run RAL_MLP_LDS=65536
run RAL_MLP_LDS=100000
run RAL_ATTN_SPLIT=1
run RAL_ATTN_SPLIT=4
run RAL_ATTN_FWD_V=64:128
run RAL_ATTN_FWD_V=32:256
run RAL_ATTN_BWD_V=32:64
run RAL_FUSE_DW=16
run RAL_QKV_BF16=0
run RAL_QKV_BF16=128
run RAL_X=default
