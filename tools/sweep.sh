#!/bin/bash
# Environment-knob sweep on the GPU box: each line is one bench.py run (30 steps) with the given variables.
#   gpurun -- 'bash tools/sweep.sh'
run() { echo "=== $*"; env "$@" python bench.py --steps 30 --warmup 3 --no-cpu --no-infer 2>&1 | grep -E "value" | sed -e 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/windows\/s \1  ms \2/'; }
run RAL_GRID_QKVB=256 RAL_GRID_RESB=256 RAL_GRID_MLPB=512
run RAL_GRID_QKVB=256 RAL_GRID_RESB=256 RAL_GRID_MLPB=1024
run RAL_GRID_QKVB=128 RAL_GRID_RESB=256 RAL_GRID_MLPB=512
run RAL_GRID_QKVB=384 RAL_GRID_RESB=256 RAL_GRID_MLPB=512
run RAL_GRID_QKVB=256 RAL_GRID_RESB=128 RAL_GRID_MLPB=512
run RAL_GRID_QKVB=256 RAL_GRID_RESB=384 RAL_GRID_MLPB=512
run RAL_GRID_QKVB=256 RAL_GRID_RESB=256 RAL_GRID_MLPB=384
run RAL_GRID_QKVB=256 RAL_GRID_RESB=256 RAL_GRID_MLPB=640
