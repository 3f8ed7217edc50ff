run() { echo "=== $*"; env "$@" python bench.py --steps 4 --warmup 2 --no-cpu --kinds 2>&1 | grep -E "^  dw|value" | cut -c1-120; }
run RAL_DW_KSPLIT=256,256,256,128,64
run RAL_DW_KSPLIT=1024,1024,1024,128,64
run RAL_DW_KSPLIT=1024,1024,1024,128,64 RAL_DW_LDS=24000
run RAL_DW_KSPLIT=2048,2048,1024,256,128 RAL_DW_LDS=24000
run RAL_DW_KSPLIT=512,512,512,256,128 RAL_DW_LDS=32000
