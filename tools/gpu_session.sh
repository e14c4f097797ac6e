#!/bin/bash
# What one round-end GPU call runs (from the repo root through gpurun): the GPU test suite, then tools/collect_profiles.sh (per-stage breakdown,
# kernel trace, counter passes, gather ceiling, final bench line).  Copy what is judged from gpurun_out/ into profiles/ afterwards.
mkdir -p gpurun_out/${ROUND:-r05}
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${ROUND:-r05}/pytest_all.txt 2>&1; grep -n "passed\|failed" gpurun_out/${ROUND:-r05}/pytest_all.txt | tail -2
ROUND=${ROUND:-r05} timeout 1500 bash tools/collect_profiles.sh > gpurun_out/${ROUND:-r05}/collect.log 2>&1; tail -3 gpurun_out/${ROUND:-r05}/collect.log
