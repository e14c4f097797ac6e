L=/root/repo/mvsformer_amd
for rep in 1 2 3; do
for v in base "" l2t; do
  echo "## variant '${v:-interior+l1table}' $rep"
  if [ -z "$v" ]; then lib=$L/libmvs_hip.so; else lib=$L/libmvs_hip_$v.so; fi
  MVS_HIP_LIB=$lib timeout 300 python tools/bench_vis.py 2>&1 | grep "stage[34]" | sed 's/| valu.*x3 \([0-9.]* ms\).*max diff vs valu\(.*\)/| vis x3 \1 \2/'
done
done
