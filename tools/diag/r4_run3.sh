cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export ATTN_LEVELS=${LV:-3} ATTN_ONLY=bwd ATTN_NOCHECK=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r4/pmcA -- python3 tools/attn_bench.py 2048 3 > gpurun_out/r4/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r4/pmcB -- python3 tools/attn_bench.py 2048 3 > gpurun_out/r4/pmcB.log 2>&1
python3 tools/diag/pmc_sum.py gpurun_out/r4/pmcA attn_bwd; python3 tools/diag/pmc_sum.py gpurun_out/r4/pmcB attn_bwd
rm -rf gpurun_out/r4/pmcA gpurun_out/r4/pmcB
