#!/bin/bash
# Component ablation of the step's kernels: serialised rocprofv3 kernel statistics of the default library next to the
# diagnostic variants that drop one cost (tools/diag/libralenet_<variant>.so, built with
#   make -C ecg_denoise_amd/csrc VARIANT=nomfma EXTRA=-DRAL_NOMFMA     and  VARIANT=nogelu EXTRA=-DRAL_NOGELU).
# Prints average us per kernel: base | variant1 | variant2 ...     gpurun -- 'bash tools/diag/ablate.sh nomfma nogelu'
cd /tmp && export TMPDIR=/tmp
export RAL_LANES=1 RAL_NO_SIDE_STREAM=1
run() { # name, lib
  D=/tmp/abl_$1; rm -rf $D
  RAL_LIB_PATH=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 "$GRAFT_REPO_ROOT/bench.py" --min-seconds 0 --steps 5 --warmup 2 --no-cpu --no-infer > $D.log 2>&1
  find $D -name "*kernel_stats.csv" | head -1
}
F0=$(run base "$GRAFT_REPO_ROOT/ecg_denoise_amd/libralenet.so")
FS=""
for v in "$@"; do FS="$FS $(run $v $GRAFT_REPO_ROOT/tools/diag/libralenet_$v.so)"; done
python3 - $F0 $FS <<'PY'
import csv, sys, re
def load(f):
    return {re.sub(r"\(.*", "", r["Name"]).replace("void ", ""): (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(f))}
tabs = [load(f) for f in sys.argv[1:]]
rows = sorted(tabs[0].items(), key=lambda kv: -kv[1][0] * kv[1][1])
for k, (us, n) in rows[:int(__import__("os").environ.get("ABL_ROWS", "45"))]:
    print(f"{k[:44]:44s} x{n:4d} {us:8.1f} | " + " | ".join(f"{t[k][0]:8.1f}" if k in t else "       -" for t in tabs[1:]))
PY
