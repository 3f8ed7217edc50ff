cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/uks -- python3 tools/unet_bench.py > gpurun_out/r4/uks.log 2>&1
f=$(ls gpurun_out/r4/uks/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: r["Name"]):
    n = r["Name"]
    if "k_unet" in n or "k_loss" in n or "k_adam" in n or "fill" in n.lower() or "elementwise" in n:
        print(f"{float(r['AverageNs'])/1e3:7.1f} us x {int(r['Calls']):5d}  {n[:80]}")
PY
rm -rf gpurun_out/r4/uks
