set -x
mkdir -p gpurun_out/r04
L=/root/repo/mvsformer_amd
timeout 900 python -m pytest tests/test_hip_x3.py tests/test_hip_parity.py -x -q -m gpu -k "vis or x3 or golden or cascade or stage" > gpurun_out/r04/pytest_split.txt 2>&1; tail -4 gpurun_out/r04/pytest_split.txt
rm -f gpurun_out/r04/bench_split.txt
for rep in 1 2; do
for v in base ""; do
  echo "## variant '${v:-new}' $rep" >> gpurun_out/r04/bench_split.txt
  if [ -z "$v" ]; then lib=$L/libmvs_hip.so; else lib=$L/libmvs_hip_$v.so; fi
  MVS_HIP_LIB=$lib timeout 300 python tools/bench_vis.py 2>&1 | grep stage | sed 's/| valu.*x3 \([0-9.]* ms\).*/| vis x3 \1/' >> gpurun_out/r04/bench_split.txt
  MVS_HIP_LIB=$lib timeout 300 python tools/bench_x3.py --stages 3,4 --out r04/tmp_x3.txt > /dev/null 2>&1
  sed 's/| direct.*x3 \([0-9.]* ms\).*/| x3 \1/; s/| fp32 tail.*x3 tail \([0-9.]* ms\).*/| x3 tail \1/' gpurun_out/r04/tmp_x3.txt >> gpurun_out/r04/bench_split.txt
done
done
cat gpurun_out/r04/bench_split.txt
