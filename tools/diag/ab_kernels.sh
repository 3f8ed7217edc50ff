# same-box serialised kernel times of two library builds for kernels matching a pattern:
#   bash tools/diag/ab_kernels.sh <variant .so> <pattern> [rounds]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
for i in $(seq 1 ${3:-2}); do
for lib in "" "$1"; do
  echo "== ${lib:-default}"
  ( export RAL_LANES=1 RAL_NO_SIDE_STREAM=1 RAL_LIB_PATH=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5/abk -- python3 bench.py --min-seconds 0 --steps 5 --warmup 2 --no-cpu --no-infer --no-fp32 > gpurun_out/r5/abk.log 2>&1
  f=$(ls gpurun_out/r5/abk/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$2" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: r["Name"]):
    if re.search(sys.argv[2], r["Name"]):
        print(f"{float(r['AverageNs'])/1e3:7.1f} us x {int(r['Calls']):5d}  {r['Name'][:80]}")
PY
  rm -rf gpurun_out/r5/abk )
done; done
