set -x
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_hip_x3.py -x -q -m gpu -k "deconv" 2>&1 | tail -3
for i in 1 2; do
timeout 200 python tools/bench_x3.py --stages 3,4 --only conv7,conv9 --out r04/bench_deconv_pf_$i.txt > /dev/null
MVS_HIP_LIB=$PWD/mvsformer_amd/libmvs_hip_nopf.so timeout 200 python tools/bench_x3.py --stages 3,4 --only conv7,conv9 --out r04/bench_deconv_nopf_$i.txt > /dev/null
done
cat gpurun_out/r04/bench_deconv_pf_*.txt; echo ---; cat gpurun_out/r04/bench_deconv_nopf_*.txt
