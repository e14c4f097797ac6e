set -x
mkdir -p gpurun_out/r04
for k in 1536 768 512 384 1536; do
MVS_X3_SEG_BLOCKS=$k timeout 200 python tools/bench_x3.py --stages 3,4 --only conv1,conv2,conv3,conv4,conv5,conv6,conv7,conv9 --out r04/bench_segb_$k.txt > /dev/null
echo "== seg_blocks $k"; sed -e 's/ | x3 vs.*//' -e 's/ | max diff.*//' -e 's/ GF | direct [0-9.]* ms  wino [0-9a-z.]* ms / /' -e 's/ GF | direct [0-9.]* ms / /' gpurun_out/r04/bench_segb_$k.txt
done
