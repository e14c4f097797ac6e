set -x
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_all2.txt 2>&1
tail -6 gpurun_out/r04/pytest_all2.txt
timeout 300 python tools/stage_breakdown.py > gpurun_out/r04/stage_breakdown.txt 2>&1
grep -E "^stage|all stages" gpurun_out/r04/stage_breakdown.txt
timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-other-configs > gpurun_out/r04/bench_b.json 2> gpurun_out/r04/bench_b.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/bench_b.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_repeats','latency_ms_single_stream','kernel_ms_sum')})
PY
