mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_all.txt 2>&1; grep -n "passed\|failed" gpurun_out/r04/pytest_all.txt | tail -2
ROUND=r04 timeout 1500 bash tools/collect_profiles.sh > gpurun_out/r04/collect.log 2>&1; tail -3 gpurun_out/r04/collect.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_final_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_repeats','latency_ms_single_stream','kernel_ms_sum','max_rel_depth_err') if k in d})
print(d.get('roofline')); print(d['roofline_cost_volume']['frac'], d['roofline_cost_volume']['frac_of_gather_bound']); print(d.get('cpu_baseline')); print(d.get('traffic_source')); print(d.get('other_configs'))
PY
