export ATTN_LEVELS=2,3 ATTN_ONLY=bwd ATTN_NOCHECK=1
P='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if "N" in d: print("   N=%d Len=%d bwd %.1f us frac %.3f" % (d["N"], d["Len"], d["bwd_us"], d["bwd_frac"]))'
echo "default"; python tools/attn_bench.py 2>/dev/null | python -c "$P"
for v in noatom notabread; do echo "variant $v"; RAL_LIB_PATH=$PWD/tools/diag/libralenet_$v.so python tools/attn_bench.py 2>/dev/null | python -c "$P"; done
