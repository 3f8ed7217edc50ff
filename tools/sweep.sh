run() { echo "=== $*"; for i in 1 2; do env "$@" python bench.py --steps 20 --warmup 3 --no-cpu 2>&1 | grep -E "value" | cut -c1-110; done; }
run RAL_DW_PRIO=2
run RAL_DW_PRIO=2 RAL_LIB_PATH=tools/diag/libralenet_s64w256.so
run RAL_DW_PRIO=0
