python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -5
python bench.py --kinds --no-cpu --no-infer --steps 20 2>&1 | grep -v amdgpu | cut -c1-400
RAL_MLP_FWD_W=0 python bench.py --kinds --no-cpu --no-infer --steps 20 2>&1 | grep -v amdgpu | cut -c1-200
