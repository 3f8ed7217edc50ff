cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for v in 3 4; do
echo "mlp_fwd_w=$v"
( export RAL_MLP_FWD_W=$v RAL_LANES=1 RAL_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/mf -- python3 bench.py --min-seconds 0 --steps 5 --warmup 2 --no-cpu --no-infer --no-fp32 > gpurun_out/r4/mf.log 2>&1
f=$(ls gpurun_out/r4/mf/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: r["Name"]):
    n = r["Name"]
    if "k_mlp_fwd" in n:
        print(f"{float(r['AverageNs'])/1e3:7.1f} us x {int(r['Calls']):5d}  {n[:70]}")
PY
rm -rf gpurun_out/r4/mf )
python3 bench.py --opt mlp_fwd_w=$v --steps 40 --warmup 5 --no-cpu --no-infer --no-fp32 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
done
