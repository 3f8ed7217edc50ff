#!/bin/bash
# Per-kernel statistics of the bench step for kernels matching a pattern, optionally with extra environment settings:
#   gpurun -- 'bash tools/diag/prof_kernel.sh "qkv_fwd|split" --opt f16_split=0'
PAT=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
D=/tmp/prof_kernel_$$
export RAL_LANES=1 RAL_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 "$GRAFT_REPO_ROOT/bench.py" --min-seconds 0 --steps 5 --warmup 2 --no-cpu --no-infer > $D.log 2>&1
f=$(find $D -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { tail -5 $D.log; exit 1; }
head -1 "$f" | cut -c1-200
grep -E "$PAT" "$f" | cut -c1-260
