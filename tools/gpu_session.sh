mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_all.txt 2>&1; tail -4 gpurun_out/r04/pytest_all.txt
ROUND=r04 timeout 1500 bash tools/collect_profiles.sh > gpurun_out/r04/collect.log 2>&1; tail -5 gpurun_out/r04/collect.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_final_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','latency_ms_single_stream','kernel_ms_sum') if k in d})
print(d.get('roofline')); print(d.get('roofline_cost_volume')); print(d.get('cpu_baseline'))
PY
