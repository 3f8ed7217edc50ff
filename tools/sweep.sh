run() { echo "=== $*"; for i in 1; do env "$@" python bench.py --steps 20 --warmup 3 --no-cpu --kinds 2>&1 | grep -E "mlp_bwd|^  dw|value" | cut -c1-110; done; }
run RAL_FUSE_DW=0
run RAL_FUSE_DW=8
run RAL_FUSE_DW=16
run RAL_FUSE_DW=32
