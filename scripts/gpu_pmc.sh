cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
grep -c . $R/gpurun_out/counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python $R/scripts/pmc_conv.py "rpn 3x3" fwd > $R/gpurun_out/pmc_$tag.log 2>&1
  ls $R/gpurun_out/pmc_$tag
done
