# usage: kstats.sh <tag> [name pattern]: rocprofv3 kernel stats of the serialised step (one lane, no side stream)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export RAL_LANES=1 RAL_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/ks_$1 -- python3 bench.py --min-seconds 0 --steps 5 --warmup 2 --no-cpu --no-infer --no-fp32 > gpurun_out/r4/ks_$1.log 2>&1
f=$(ls gpurun_out/r4/ks_$1/*/*kernel_stats.csv | head -1)
python3 - "$f" "${2:-}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
for r in rows:
    n = r["Name"]
    if pat and pat not in n: continue
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x {int(r['Calls']):5d} = {float(r['TotalDurationNs'])/1e6:8.2f} ms  {n[:110]}")
PY
cp "$f" gpurun_out/r4/ks_$1.csv; rm -rf gpurun_out/r4/ks_$1
