# LDS pressure per kernel of the serialised training step (one --pmc pass, kernel trace only):
#   gpurun -- 'bash tools/diag/pmc_lds_step.sh tag'   -> gpurun_out/pmc_lds_<tag>.txt  (per kernel: LDS-array busy share of the CU cycles,
#   bank-conflict share, waves stalled at LDS issue, vector-issue busy)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-x}; O=gpurun_out/pmc_lds_$T
rm -rf $O.d
export RAL_LANES=1 RAL_NO_SIDE_STREAM=1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O.d -- python3 bench.py --min-seconds 0 --steps 2 --warmup 1 --no-cpu --no-infer --no-fp32 > $O.log 2>&1
python3 - $O.d > $O.txt <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(collections.Counter); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:56]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
rows = []
for k, c in acc.items():
    L = n[(k, "GRBM_GUI_ACTIVE")]
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
    if cyc <= 0: continue
    rows.append((cyc, k, L, c))
rows.sort(reverse=True)
print(f"{'kernel':58s} {'launches':>8s} {'Mcyc tot':>9s} {'LDS busy':>9s} {'conflict':>9s} {'LDSstall':>9s} {'VALU busy':>9s}")
for cyc, k, L, c in rows[:40]:
    lds = c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256.0)
    print(f"{k:58s} {L:8d} {cyc/1e6:9.2f} {lds:9.3f} {c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):9.3f} "
          f"{c['SQ_WAIT_INST_LDS']/max(c['SQ_WAVE_CYCLES'],1):9.3f} {4*c['SQ_ACTIVE_INST_VALU']/(cyc*1024.0):9.3f}")
PY
rm -rf $O.d
cat $O.txt
