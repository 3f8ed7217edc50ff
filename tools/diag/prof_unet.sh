cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/unet_prof -- python3 tools/unet_bench.py > gpurun_out/unet_prof.log 2>&1
f=$(ls gpurun_out/unet_prof/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-160
