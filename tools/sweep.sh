#!/bin/bash
run() { echo "=== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu --infer 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('infer_windows_per_s'))"; }
run RAL_LANES=1
run RAL_LANES=2
run RAL_LANES=4
run RAL_LANES=2 RAL_NO_SIDE_STREAM=1
