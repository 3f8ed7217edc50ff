export ATTN_ONLY=bwd ATTN_LEVELS=2,3,4 ATTN_NOCHECK=1
P='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if "N" in d: print("   N=%d Len=%d bwd %.1f us frac %.3f" % (d["N"], d["Len"], d["bwd_us"], d["bwd_frac"]))'
for w in 4 2 3 1; do echo "waves $w"; RAL_ATTNW_WAVES=$w python tools/attn_bench.py 2>/dev/null | python -c "$P"; done
echo default; python tools/attn_bench.py 2>/dev/null | python -c "$P"
