# serialised per-launch times of the MLP kernels (forward and backward) with the library given in RAL_LIB_PATH (or the default)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
( export RAL_LANES=1 RAL_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/ms -- python3 bench.py --min-seconds 0 --steps 5 --warmup 2 --no-cpu --no-infer --no-fp32 > gpurun_out/r4/ms.log 2>&1
f=$(ls gpurun_out/r4/ms/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: r["Name"]):
    n = r["Name"]
    if "k_mlp_" in n:
        print(f"{float(r['AverageNs'])/1e3:7.1f} us x {int(r['Calls']):5d}  {n[:70]}")
PY
rm -rf gpurun_out/r4/ms )
python3 bench.py --steps 40 --warmup 5 --no-cpu --no-infer --no-fp32 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
