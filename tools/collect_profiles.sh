#!/bin/bash
# Round profile collection on the GPU box (one gpurun call): rocprofv3 kernel statistics of the bench.py step in the
# default and in the serialised schedule, the HBM-traffic and SQ counter passes (every --pmc pass is its own run with
# --kernel-trace only), the same for the U-Net forward, plus the attention and VALU micro-benchmarks.
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r03'
# Everything lands under gpurun_out/<round>/; tools/summarise_profiles.py turns it into profiles/<round>_*.
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R
rm -rf "$O"; mkdir -p "$O"
BENCH="python3 bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu --no-infer --no-fp32"   # (exactly 5 timed steps: a 2 s region is a 60 MB trace)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_default -- $BENCH > $O/ks_default.log 2>&1
export RAL_LANES=1 RAL_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_serial -- $BENCH > $O/ks_serial.log 2>&1
B1="python3 bench.py --steps 1 --warmup 1 --min-seconds 0 --no-cpu --no-infer --no-fp32"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- $B1 > $O/pmc_sq.log 2>&1
unset RAL_LANES RAL_NO_SIDE_STREAM
# U-Net forward: fused (default) and stage by stage
U="python3 tools/unet_bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_ks -- $U > $O/unet_ks.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/unet_fetch -- $U > $O/unet_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/unet_write -- $U > $O/unet_write.log 2>&1
export RAL_TOOL_OPTIONS=unet_fused=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_staged_ks -- $U > $O/unet_staged_ks.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/unet_staged_fetch -- $U > $O/unet_staged_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/unet_staged_write -- $U > $O/unet_staged_write.log 2>&1
unset RAL_TOOL_OPTIONS
python3 tools/attn_bench.py > $O/attn_bench.log 2>&1
RAL_TOOL_OPTIONS=attn_f16=0 python3 tools/attn_bench.py > $O/attn_bench_fp32.log 2>&1
python3 tools/diag/step_timeline_events.py $O/step_timeline_events.txt > /dev/null 2>&1
RAL_LANES=1 RAL_NO_SIDE_STREAM=1 python3 tools/diag/step_timeline_events.py $O/step_timeline_events_serial.txt > /dev/null 2>&1
[ -x tools/diag/valu_probe ] && ./tools/diag/valu_probe > $O/valu_probe.log 2>&1
python3 bench.py --steps 50 --warmup 5 --kinds > $O/bench.json 2> $O/bench_kinds.log
python3 bench.py --config newrale --no-cpu > $O/bench_newrale.json 2> $O/bench_newrale.err
python3 bench.py --config unet --no-cpu > $O/bench_unet.json 2> $O/bench_unet.err
python3 tools/config_bench.py > $O/config_bench.log 2>&1
python3 tools/baselines_bench.py > $O/baselines_bench.log 2>&1
# U-Net TRAIN step: kernel trace (timeline of one step) and statistics
rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_train_ks -- python3 tools/unet_bench.py > $O/unet_train_ks.log 2>&1
python3 tools/diag/unet_timeline.py $O/unet_train_ks 20 > $O/unet_train_timeline.txt 2>&1
# data-parallel step, two ranks sharing this GPU (gloo on device tensors): kernel + memory-copy trace of both ranks - does the
# early gradient bucket's reduction start before the backward pass has ended?
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/dp2_trace -- python3 bench.py --gpus 2 --steps 4 --warmup 2 --min-seconds 0 --batch 1024 --no-cpu --no-infer --test-backend gloo --test-share-gpu > $O/dp2_trace.log 2>&1
python3 tools/diag/dp_overlap.py $O/dp2_trace > $O/dp2_overlap.txt 2>&1
# the CPU baseline at the bench batch (SURVEY 8d ii; ~6 minutes of host time: SKIP_CPU_BIG=1 leaves it out of a re-collection)
[ -z "$SKIP_CPU_BIG" ] && python3 tools/cpu_baseline_big.py 2048 8 32 64 > $O/cpu_baseline_b2048.json 2> $O/cpu_baseline_b2048.err
# keep only the csv summaries (the merged directory is capped at 64 MiB)
find $O -name "*agent_info.csv" -delete
du -sh $O
