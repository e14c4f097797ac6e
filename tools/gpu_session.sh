set -x
mkdir -p gpurun_out/r04
( timeout 200 tools/probe/pk_mfma_race_fence 3000 ) > gpurun_out/r04/pk_mfma_race_fence.txt 2>&1
timeout 900 python -m pytest tests/test_hip_x3.py -x -q -m gpu > gpurun_out/r04/pytest_x3.txt 2>&1
tail -15 gpurun_out/r04/pytest_x3.txt
X3_ABLATION=1 timeout 300 python tools/bench_x3.py --stages 3,4 --only conv2,conv4,conv6,tail --out r04/bench_x3_db1.txt
MVS_X3_DB=0 timeout 300 python tools/bench_x3.py --stages 3,4 --only conv2,conv4,conv6 --out r04/bench_x3_db0.txt
head -12 gpurun_out/r04/pk_mfma_race_fence.txt
