#!/bin/bash
# Kernel trace of the graph-replayed training step: per-kernel stats of the last 5 replays -> gpurun_out/r05/train_trace_$1.csv
T=${1:-x}
REPO=$(pwd)
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_tt -o t -- python $REPO/bench_train.py --steps 10 --warmup 2 > $REPO/gpurun_out/r05/tt_$T.json 2>/dev/null)
DBT=$(find gpurun_out/prof_tt -name "*.db" | head -1)
NK=$(python -c "import json; print(json.load(open('gpurun_out/r05/tt_$T.json'))['launches_per_step']['kernel']*5)")
python tools/rocpd_stats.py $DBT $NK > gpurun_out/r05/train_trace_$T.csv
rm -rf gpurun_out/prof_tt
python - <<PY
import csv
rows=list(csv.reader(open('gpurun_out/r05/train_trace_$T.csv')))[1:]
tot=sum(int(r[2]) for r in rows)/5e6
print("sum of kernel time per step %.3f ms"%tot)
for r in rows[:28]: print("%-86s %4d %8.3f ms/step avg %7.1f us"%(r[0][:86], int(r[1])//5, int(r[2])/5e6, float(r[3])/1e3))
PY
