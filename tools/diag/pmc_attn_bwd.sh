# SQ counters of the attention BACKWARD kernels per level (two --pmc passes, each its own run with --kernel-trace only).
#   gpurun -- 'bash tools/diag/pmc_attn_bwd.sh [levels, e.g. 2,3,4] [tag]'     -> gpurun_out/pmc_attnb_<tag>.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ATTN_LEVELS=${1:-2,3,4} ATTN_ONLY=bwd
T=${2:-x}
O=gpurun_out/pmc_attnb_$T
rm -rf $O.a $O.b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O.a -- python3 tools/attn_bench.py 2048 3 > $O.a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O.b -- python3 tools/attn_bench.py 2048 3 > $O.b.log 2>&1
(python3 tools/diag/pmc_sum.py $O.a attn_bwd; python3 tools/diag/pmc_sum.py $O.b attn_bwd) > $O.txt 2>&1
rm -rf $O.a $O.b
cat $O.txt
