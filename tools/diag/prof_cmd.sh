#!/bin/bash
# rocprofv3 kernel statistics of an arbitrary python tool: gpurun -- 'bash tools/diag/prof_cmd.sh tools/unet_bench.py'
cd /tmp && export TMPDIR=/tmp
D=/tmp/prof_cmd_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 "$GRAFT_REPO_ROOT/$1" > $D.log 2>&1
tail -2 $D.log | cut -c1-400
f=$(find $D -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && exit 1
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print(f'{float(r["TotalDurationNs"]) / tot * 100:5.1f}% {float(r["AverageNs"]) / 1e3:8.1f}us x{int(r["Calls"]):5d}  {r["Name"][:120]}')
PY
