#!/bin/bash
# Round-end evidence on the GPU box (run from the repo root through gpurun): gather ceiling, per-stage breakdown, kernel trace of the bench,
# HBM traffic counters (separate --pmc passes), final bench line.  Everything lands in gpurun_out/, copy what is judged into profiles/.
set -x
export MVS_HEAD=${MVS_HEAD:-unknown}
ROUND=${ROUND:-r06}
mkdir -p gpurun_out
REPO=$(pwd)
python tools/gather_bound.py > gpurun_out/gather_bound.log 2>&1
python tools/stage_breakdown.py > /dev/null 2>&1
python tools/bench_sweeps.py --hyps cascade > /dev/null 2>&1
python tools/bench_x3.py --stages 3,4 > /dev/null 2>&1
python tools/bench_small.py > /dev/null 2>&1
python tools/bench_vis.py > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/prof_${ROUND} $REPO/gpurun_out/pmc_${ROUND}
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${ROUND} -o ${ROUND} -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --streams 1 > $REPO/gpurun_out/bench_under_rocprof.json 2> $REPO/gpurun_out/rocprof_trace.log
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $REPO/gpurun_out/pmc_${ROUND}/fetch -- python $REPO/tools/prof_traffic.py > $REPO/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $REPO/gpurun_out/pmc_${ROUND}/write -- python $REPO/tools/prof_traffic.py > $REPO/gpurun_out/pmc_write.log 2>&1
# SQ counters of the stage-4 regularizer + visibility CNN (two passes of 8 counters each; --pmc never together with a trace domain other than kernel-trace)
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $pass | cut -d" " -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $REPO/gpurun_out/pmc_${ROUND}/sq_$tag -- python $REPO/tools/prof_cv.py --stage 4 --what vis,reg > $REPO/gpurun_out/pmc_sq_$tag.log 2>&1
done
cd $REPO
python tools/pmc_table.py gpurun_out/pmc_${ROUND}/sq_SQ_WAVES gpurun_out/pmc_${ROUND}/sq_SQ_ACTIVE_INST_ANY > gpurun_out/${ROUND}_pmc_reg_stage4.txt 2> gpurun_out/pmc_table.err
DB=$(find gpurun_out/prof_${ROUND} -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > gpurun_out/${ROUND}_kernel_stats.csv 2> gpurun_out/rocpd_stats.err
python tools/pmc_traffic.py gpurun_out/pmc_${ROUND}/fetch gpurun_out/pmc_${ROUND}/write gpurun_out/traffic_by_kernel.json > gpurun_out/pmc_traffic.log 2>&1
cp gpurun_out/traffic_by_kernel.json profiles/traffic_by_kernel.json; cp gpurun_out/gather_bound.json profiles/gather_bound.json; cp gpurun_out/gather_bound.txt gpurun_out/${ROUND}_gather_bound.txt   # the final bench line below reads them
rm -rf gpurun_out/pmc_${ROUND} gpurun_out/prof_${ROUND}
# config 3 (training): the graph-replayed step, its per-kernel table (eager events) and a kernel trace of the replays
python bench_train.py --steps 30 --warmup 3 > gpurun_out/${ROUND}_bench_train.json 2> gpurun_out/${ROUND}_bench_train.err
python tools/prof_train.py --bf16 > gpurun_out/${ROUND}_train_kernels_eager.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_train_${ROUND} -o t -- python $REPO/bench_train.py --steps 10 --warmup 2 > $REPO/gpurun_out/train_under_rocprof.json 2>/dev/null)
DBT=$(find gpurun_out/prof_train_${ROUND} -name "*.db" | head -1)
NK=$(python -c "import json; print(json.load(open('gpurun_out/train_under_rocprof.json'))['launches_per_step']['kernel']*5)")
python tools/rocpd_stats.py $DBT $NK > gpurun_out/${ROUND}_train_kernel_stats.csv 2>> gpurun_out/rocpd_stats.err
rm -rf gpurun_out/prof_train_${ROUND}
# the same step under DistributedDataParallel + SyncBatchNorm over RCCL (one rank), captured whole; the ViT branch's per-kernel table
python bench_train.py --force-ddp --steps 30 --warmup 3 > gpurun_out/${ROUND}_bench_train_ddp.json 2> gpurun_out/${ROUND}_bench_train_ddp.err
python tools/prof_vit.py > gpurun_out/${ROUND}_prof_vit.txt 2>&1
python tools/bench_x3p.py > gpurun_out/${ROUND}_bench_x3p.txt 2>&1
tools/sweep_x3p.sh > gpurun_out/${ROUND}_sweep_x3p.txt 2>&1
tools/prof_vit_trace.sh > /dev/null 2>&1          # -> gpurun_out/${ROUND}_vit_kernel_stats.csv (kernel trace of the branch, ATen glue included)
python tools/exp_small_conv.py > gpurun_out/${ROUND}_bf16_small_conv.txt 2>&1
# the FPN (before the path): encoder / decoder per kernel with the split-form full-resolution layers and with the fp32-MFMA kernels, the last level's three forms, its SQ counters
python tools/bench_fpn.py --iters 20 2>/dev/null > gpurun_out/${ROUND}_fpn_encoder_decoder.json
MVS_FPN_X3=0 python tools/bench_fpn.py --iters 20 2>/dev/null > gpurun_out/${ROUND}_fpn_encoder_decoder_fp32mfma.json
python tools/bench_fpn_level.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${ROUND}_bench_fpn_level.txt
bash tools/pmc_fpn.sh; cp gpurun_out/pmc_fpn_level.txt gpurun_out/${ROUND}_pmc_fpn_level.txt
python bench.py --steps 200 --warmup 10 > gpurun_out/${ROUND}_final_bench.json 2> gpurun_out/${ROUND}_final_bench.err
tail -c 400 gpurun_out/${ROUND}_final_bench.err
ls -la gpurun_out | tail -20
