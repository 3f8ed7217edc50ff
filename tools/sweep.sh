#!/bin/bash
run() { echo "=== $*"; env "$@" python bench.py --steps 8 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run RAL_DW_KSPLIT=512,512,256,128,64
run RAL_DW_KSPLIT=256,256,128,64,32
run RAL_DW_KSPLIT=256,256,256,128,64
run RAL_DW_KSPLIT=512,512,256,64,32
run RAL_DW_KSPLIT=128,128,128,64,32
