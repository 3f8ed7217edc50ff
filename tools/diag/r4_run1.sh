set -x
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_attention.py -x -q 2>&1 | tail -15 > gpurun_out/r4/attn_test_w1.txt
RAL_ATTN_BWD_W=0 python tools/attn_bench.py > gpurun_out/r4/attn_bench_w0.jsonl 2>&1
RAL_ATTN_BWD_W=1 python tools/attn_bench.py > gpurun_out/r4/attn_bench_w1.jsonl 2>&1
RAL_ATTN_BWD_W=1 ATTN_NOTABLE=1 ATTN_LEVELS=2,3 python tools/attn_bench.py > gpurun_out/r4/attn_bench_w1_notab.jsonl 2>&1
RAL_ATTN_BWD_W=0 ATTN_NOTABLE=1 ATTN_LEVELS=2,3 python tools/attn_bench.py > gpurun_out/r4/attn_bench_w0_notab.jsonl 2>&1
cat gpurun_out/r4/*.txt gpurun_out/r4/*.jsonl
