# same-box step-time A/B of library builds (and / or switch settings), interleaved over several rounds:
#   bash tools/diag/ab_step.sh [rounds] <spec> <spec> ...    spec = "<variant .so or ->[:key=value,...]"
cd $GRAFT_REPO_ROOT
R=$1; shift
for rep in $(seq 1 $R); do
for sp in "$@"; do
  lib=${sp%%:*}; o=""; [ "$sp" != "$lib" ] && o=${sp#*:}
  [ "$lib" = "-" ] && lib=""
  a=""; for kv in $(echo $o | tr ',' ' '); do a="$a --opt $kv"; done
  printf "%-60s" "$sp"
  RAL_LIB_PATH=$lib python3 bench.py --steps 30 --warmup 5 --no-cpu --no-infer --no-fp32 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['median_ms_per_step_hipevent'])"
done; done
