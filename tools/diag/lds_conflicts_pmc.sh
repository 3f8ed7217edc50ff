cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4/ldspmc
rm -rf $O; mkdir -p $O
export RAL_LANES=1 RAL_NO_SIDE_STREAM=1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python3 bench.py --min-seconds 0 --steps 1 --warmup 1 --no-cpu --no-infer --no-fp32 > $O/a.log 2>&1
python3 - $O/a <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(collections.Counter); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
rows = []
for k, c in acc.items():
    L = n[(k, "GRBM_GUI_ACTIVE")] or 1
    gui = c["GRBM_GUI_ACTIVE"] / 8.0   # summed over 8 XCDs
    rows.append((c["SQ_LDS_BANK_CONFLICT"], k, L, gui / L, c["SQ_LDS_IDX_ACTIVE"], c["SQ_INSTS_LDS"]))
for bc, k, L, cyc, idx, insts in sorted(rows, reverse=True)[:40]:
    print(f"{k:70s} x{L:4d} cyc/launch {cyc:9.0f}  conflict/active {bc / max(idx, 1):5.2f}  lds_active/(cyc*256CU) {idx / max(cyc * L * 256, 1):5.2f}  conflict/(cyc*256) {bc / max(cyc * L * 256, 1):5.2f}")
PY
rm -rf $O/a
