mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_attention.py -q -x 2>&1 | tail -15
RAL_ATTN_F16=0 python -m pytest tests/test_gpu_attention.py -q 2>&1 | tail -5
P='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if "N" in d: print("   N=%d Len=%d fwd %.1f bwd %.1f us frac %.3f %.3f err %.1e" % (d["N"], d["Len"], d["fwd_us"], d["bwd_us"], d["fwd_frac"], d["bwd_frac"], d["max_rel_err"]))
    else: print("   ", d)'
echo "new f16"; python tools/attn_bench.py 2>/dev/null | python -c "$P"
