mkdir -p gpurun_out/r04
L=/root/repo/mvsformer_amd
timeout 600 python -m pytest tests/test_hip_x3.py -x -q -m gpu -k "tail or logits" 2>&1 | tail -3
for rep in 1 2 3; do
for v in noil ""; do
  if [ -z "$v" ]; then lib=$L/libmvs_hip.so; else lib=$L/libmvs_hip_$v.so; fi
  echo "## ${v:-interleaved} $rep"
  MVS_HIP_LIB=$lib timeout 300 python tools/bench_x3.py --stages 3,4 --only tail --out r04/tmp_x3.txt 2>&1 | grep tail | sed 's/| fp32 tail.*x3 tail \([0-9.]* ms\).*/| x3 tail \1/'
done
done
MVS_HIP_LIB=$L/libmvs_hip_tl.so timeout 300 python tools/x3_timeline.py 2>&1 | grep tail
