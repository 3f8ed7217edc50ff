#!/bin/bash
run() { echo "=== $*"; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu --kinds 2>&1 | grep -E "ms/step|value" | cut -c1-160; }
run RAL_DW_KSPLIT=256
run RAL_DW_KSPLIT=128
run RAL_DW_KSPLIT=64
