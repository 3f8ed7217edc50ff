#!/bin/bash
run() { echo "=== $*"; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu --kinds 2>&1 | grep -E "dw |resample_bwd|value" | cut -c1-120; }
run RAL_DW_LDS=51200
run RAL_DW_LDS=40000
run RAL_DW_LDS=80000
run RAL_DW_KSPLIT=2048,2048,2048,1024,256
run RAL_DW_KSPLIT=1024,1024,512,256,128
run RAL_DW_KSPLIT=2048,2048,1024,256,64
