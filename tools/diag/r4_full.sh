mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4/gputest.txt
cat gpurun_out/r4/gputest.txt
