export ATTN_ONLY=bwd ATTN_LEVELS=0,1 
P='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if "N" in d: print("   N=%d Len=%d bwd %.1f us frac %.3f err %.1e" % (d["N"], d["Len"], d["bwd_us"], d["bwd_frac"], d["max_rel_err"]))'
for g in 1024 2048 4096; do echo "grid $g"; RAL_GRID_ATTNH=$g python tools/attn_bench.py 2>/dev/null | python -c "$P"; done
