# same-box A/B of two builds of the library on the attention backward: bash tools/diag/ab_attn.sh <variant .so> [levels] [rounds]
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${3:-3}); do
  for lib in "" "$1"; do
    printf "%-40s" "${lib:-default}"
    RAL_LIB_PATH=$lib ATTN_LEVELS=${2:-0,1} ATTN_ONLY=bwd ATTN_NOCHECK=1 python3 tools/attn_bench.py 2>&1 | grep bwd_us | sed 's/.*"N": \([0-9]*\).*"bwd_us": \([0-9.]*\).*/N=\1: \2 us/' | tr '\n' ' '; echo
  done
done
