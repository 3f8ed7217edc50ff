export ATTN_ONLY=bwd ATTN_LEVELS=0,1 ATTN_NOCHECK=1
P='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if "N" in d: print("   N=%d Len=%d bwd %.1f us frac %.3f" % (d["N"], d["Len"], d["bwd_us"], d["bwd_frac"]))'
echo "default"; python tools/attn_bench.py 2>/dev/null | python -c "$P"
echo "lds 32K (hg 1 at N=256, 256 threads)"; RAL_ATTNH_LDS=32768 python tools/attn_bench.py 2>/dev/null | python -c "$P"
echo "lds 32K 512 threads"; RAL_ATTNH_LDS=32768 RAL_ATTNH_THREADS=512 python tools/attn_bench.py 2>/dev/null | python -c "$P"
echo "default 256 threads"; RAL_ATTNH_THREADS=256 python tools/attn_bench.py 2>/dev/null | python -c "$P"
echo "grid 2048"; RAL_GRID_ATTNH=2048 python tools/attn_bench.py 2>/dev/null | python -c "$P"
echo "lds 32K grid 4096"; RAL_ATTNH_LDS=32768 RAL_GRID_ATTNH=4096 python tools/attn_bench.py 2>/dev/null | python -c "$P"
