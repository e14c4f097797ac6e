cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $pass | cut -d" " -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $REPO/gpurun_out/pmc_fpn/sq_$tag -- python $REPO/tools/bench_fpn_level.py > $REPO/gpurun_out/pmc_fpn_$tag.log 2>&1
done
cd $REPO
python tools/pmc_table.py gpurun_out/pmc_fpn/sq_SQ_WAVES gpurun_out/pmc_fpn/sq_SQ_ACTIVE_INST_ANY > gpurun_out/pmc_fpn_level.txt 2> gpurun_out/pmc_fpn_table.err
rm -rf gpurun_out/pmc_fpn
