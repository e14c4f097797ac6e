L=/root/repo/mvsformer_amd
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_x3.py tests/test_hip_multistream.py -x -q -m gpu -k "vis or golden or cascade or stage or default" 2>&1 | tail -3
for rep in 1 2 3; do
for v in base ""; do
  echo "## variant '${v:-folded}' $rep"
  if [ -z "$v" ]; then lib=$L/libmvs_hip.so; else lib=$L/libmvs_hip_$v.so; fi
  MVS_HIP_LIB=$lib timeout 300 python tools/bench_vis.py 2>&1 | grep "stage" | sed 's/| valu.*x3 \([0-9.]* ms\).*max diff vs valu\(.*\)/| vis x3 \1 \2/'
done
done
