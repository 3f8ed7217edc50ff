run() { echo "=== $*"; for i in 1; do env "$@" python bench.py --steps 20 --warmup 3 --no-cpu 2>&1 | grep -E "value" | cut -c1-110; done; }
run RAL_DEBUG_SKIP_DW=1 RAL_LIB_PATH=tools/diag/libralenet_skipdw.so
run RAL_LIB_PATH=tools/diag/libralenet_skipdw.so
run RAL_FUSE_DW=0
run RAL_FUSE_DW=1
run RAL_LANES=1
run RAL_LANES=1 RAL_NO_SIDE_STREAM=1
