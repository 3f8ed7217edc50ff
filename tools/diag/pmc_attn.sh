cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ATTN_LEVELS=0 ATTN_ONLY=fwd
for v in 64 0; do
  export RAL_ATTN_FWD_V=$v
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_v$v -- python3 tools/attn_bench.py 2048 3 > gpurun_out/pmc_v$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc2_v$v -- python3 tools/attn_bench.py 2048 3 > gpurun_out/pmc2_v$v.log 2>&1
done
ls gpurun_out/pmc_v64/*/ | head
