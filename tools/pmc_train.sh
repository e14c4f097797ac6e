cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $pass | cut -d" " -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $R/gpurun_out/r05/pmc_tr_$tag -- python $R/tools/prof_train.py --bf16 > /dev/null 2>&1
done
cd $R
python tools/pmc_table.py gpurun_out/r05/pmc_tr_SQ_WAVES gpurun_out/r05/pmc_tr_SQ_ACTIVE_INST_ANY > gpurun_out/r05/pmc_train.txt 2>&1
rm -rf gpurun_out/r05/pmc_tr_*
