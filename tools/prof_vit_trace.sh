#!/bin/bash
# kernel trace of the ViT branch (tools/prof_vit.py) -> per-kernel stats incl. the ATen glue kernels
REPO=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/prof_vit_trace
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_vit_trace -o v -- python $REPO/tools/prof_vit.py > $REPO/gpurun_out/prof_vit_under_rocprof.txt 2>&1
cd $REPO
DB=$(find gpurun_out/prof_vit_trace -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > gpurun_out/${ROUND:-r06}_vit_kernel_stats.csv 2> gpurun_out/rocpd_stats_vit.err
head -40 gpurun_out/${ROUND:-r06}_vit_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/prof_vit_trace
