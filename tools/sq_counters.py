"""Summarise one rocprofv3 PMC pass of SQ/GRBM counters per bench.py kernel kind -> profiles/rNN_sq_counters.json.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY \
              SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
              -- python3 bench.py --steps 1 --warmup 1 --no-cpu          (RAL_LANES=1 RAL_NO_SIDE_STREAM=1)
    python tools/sq_counters.py <counter_collection.csv> profiles/r01_sq_counters.json

Units (MI355X_MICROARCH.md, constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles,
SQ_VALU_MFMA_BUSY_CYCLES counts cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Derived per kind:
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs)     matrix-pipe busy fraction
  valu_issue     = 4 x SQ_ACTIVE_INST_VALU / (kernel cycles x 1024 SIMDs)      vector-issue busy fraction (includes MFMA issue)
  wave_*         = shares of SQ_WAVE_CYCLES: executing / parked (s_waitcnt, barrier) / issue-stalled"""
import collections, csv, json, sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hbm_traffic import KIND_OF, family


def main():
    src, out = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.Counter())
    n = collections.Counter()
    for r in csv.DictReader(open(src)):
        k = KIND_OF.get(family(r["Kernel_Name"]))
        if not k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
    res = {}
    for k, c in acc.items():
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        simd = cyc * 1024.0
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        res[k] = {"launches": n[k], "kernel_cycles_per_launch": round(cyc / max(n[k], 1)),
                  "mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 4),
                  "valu_issue": round(4.0 * c["SQ_ACTIVE_INST_VALU"] / simd, 4),
                  "mfma_mops_f32_per_launch": round(c["SQ_INSTS_VALU_MFMA_MOPS_F32"] / max(n[k], 1)),
                  "wave_executing": round(c["SQ_ACTIVE_INST_ANY"] / wc, 4), "wave_parked": round(c["SQ_WAIT_ANY"] / wc, 4),
                  "wave_issue_stalled": round(c["SQ_WAIT_INST_ANY"] / wc, 4),
                  "waves_per_simd_avg": round(4.0 * c["SQ_WAVE_CYCLES"] / simd, 2)}
    json.dump({"note": __doc__.split("Units")[1].strip(), "per_kind": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print(k, v)


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
