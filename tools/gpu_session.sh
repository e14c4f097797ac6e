set -x
mkdir -p gpurun_out/r04
L=/root/repo/mvsformer_amd
for rep in 1 2; do
for v in base "" wr3 wr3mb2; do
  echo "## variant '${v:-default}' $rep" >> gpurun_out/r04/bench_x3_wring.txt
  if [ -z "$v" ]; then lib=$L/libmvs_hip.so; else lib=$L/libmvs_hip_$v.so; fi
  MVS_HIP_LIB=$lib timeout 300 python tools/bench_x3.py --stages 3,4 --only conv1,conv2,conv3,conv4,conv5,conv6 --out r04/tmp_x3.txt > /dev/null 2>&1
  sed 's/| direct.*x3 \([0-9.]* ms\).*/| x3 \1/' gpurun_out/r04/tmp_x3.txt >> gpurun_out/r04/bench_x3_wring.txt
done
done
cat gpurun_out/r04/bench_x3_wring.txt
MVS_HIP_LIB=$L/libmvs_hip_tl.so timeout 300 python tools/x3_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/x3_timeline.txt
