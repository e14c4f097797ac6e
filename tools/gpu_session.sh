mkdir -p gpurun_out/r04
MVS_X3_TILE_ROWS=12 timeout 900 python -m pytest tests/test_hip_x3.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_hip_x3.py -x -q -m gpu 2>&1 | tail -2
rm -f gpurun_out/r04/bench_x3_rows.txt
for rep in 1 2; do
for v in 16 12 auto; do
  echo "## tile rows $v ($rep)" >> gpurun_out/r04/bench_x3_rows.txt
  if [ $v = auto ]; then unset MVS_X3_TILE_ROWS; else export MVS_X3_TILE_ROWS=$v; fi
  timeout 300 python tools/bench_x3.py --stages 2,3,4 --only conv1,conv2 --out r04/tmp_x3.txt > /dev/null 2>&1
  sed 's/| direct.*x3 \([0-9.]* ms\).*/| x3 \1/' gpurun_out/r04/tmp_x3.txt >> gpurun_out/r04/bench_x3_rows.txt
done
done
cat gpurun_out/r04/bench_x3_rows.txt
