mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_attention.py -q 2>&1 | tail -5
export ATTN_ONLY=bwd
P='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if "N" in d: print("   N=%d Len=%d bwd %.1f us frac %.3f err %.1e" % (d["N"], d["Len"], d["bwd_us"], d["bwd_frac"], d["max_rel_err"]))
    else: print("   ", d)'
echo "old kernels"; RAL_ATTN_BWD_W=0 RAL_ATTN_BWD_H=0 python tools/attn_bench.py 2>/dev/null | python -c "$P"
echo "new fp32"; RAL_ATTN_F16=0 python tools/attn_bench.py 2>/dev/null | python -c "$P"
echo "new f16"; python tools/attn_bench.py 2>/dev/null | python -c "$P"
echo "new f16 grid 512"; RAL_GRID_ATTNH=512 ATTN_LEVELS=0,1 python tools/attn_bench.py 2>/dev/null | python -c "$P"
echo "new f16 grid 2048"; RAL_GRID_ATTNH=2048 ATTN_LEVELS=0,1 python tools/attn_bench.py 2>/dev/null | python -c "$P"
