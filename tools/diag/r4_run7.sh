for i in 1 2; do
python bench.py --no-cpu --no-infer --no-fp32 --steps 30 --kinds 2>&1 | grep -E "dw |ms_per_step|sum of" | sed 's/.*"ms_per_step": \([0-9.]*\).*/   shipped step \1 ms/' 
RAL_LIB_PATH=$PWD/tools/diag/libralenet_dwnoxf.so python bench.py --no-cpu --no-infer --no-fp32 --steps 30 --kinds 2>&1 | grep -E "dw |ms_per_step|sum of" | sed 's/.*"ms_per_step": \([0-9.]*\).*/   no-recompute dW step \1 ms/'
done
