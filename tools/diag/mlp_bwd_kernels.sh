cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
for V in ${@:-2 3}; do

OPT="--opt mlp_bwd_w=$V ${XOPT}"
echo "== mlp_bwd_w=$V"
( export RAL_LANES=1 RAL_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5/mb -- python3 bench.py --min-seconds 0 $OPT ${BATCH:+--batch $BATCH} --steps 5 --warmup 2 --no-cpu --no-infer --no-fp32 > gpurun_out/r5/mb.log 2>&1
f=$(ls gpurun_out/r5/mb/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: r["Name"]):
    n = r["Name"]
    if "k_mlp_bwd" in n or "k_mlpw" in n:
        print(f"{float(r['AverageNs'])/1e3:7.1f} us x {int(r['Calls']):5d}  {n[:70]}")
PY
rm -rf gpurun_out/r5/mb )
[ -n "$BATCH" ] || for i in 1 2; do python3 bench.py $OPT --steps 40 --warmup 5 --no-cpu --no-infer --no-fp32 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1; done
done
