#!/bin/bash
run() { echo "=== $*"; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu --kinds 2>&1 | grep -E "attn|value" | cut -c1-120; }
run RAL_ATTN_QT1=1
run RAL_X=1
