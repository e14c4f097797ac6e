#!/bin/bash
# A/B of the training step on the GPU box: $1 = tag; runs the training tests, then bench_train (graph) and the per-kernel table.
T=${1:-x}
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_training.py tests/test_hip_bf16.py tests/test_hip_graph.py -x -q -m gpu > gpurun_out/r05/pytest_$T.txt 2>&1
tail -2 gpurun_out/r05/pytest_$T.txt
python bench_train.py --steps 20 --warmup 3 > gpurun_out/r05/train_$T.json 2> gpurun_out/r05/train_$T.err
python tools/prof_train.py --bf16 > gpurun_out/r05/train_${T}_kernels.txt 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/r05/train_$T.json')); print('$T', d['ms_per_step'], d['launches_per_step'], d['final_loss'])
PY
