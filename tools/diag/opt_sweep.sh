# step time of bench.py under a list of switch settings, all in one call (same box): bash tools/diag/opt_sweep.sh "a=1" "b=2,c=3" ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for o in "" "$@"; do
  a=""; [ -n "$o" ] && for kv in $(echo $o | tr ',' ' '); do a="$a --opt $kv"; done
  printf "%-44s" "${o:-default}"
  python3 bench.py --steps 30 --warmup 5 --no-cpu --no-infer --no-fp32 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['median_ms_per_step_hipevent'])"
done; done
