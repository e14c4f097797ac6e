set -x
MVS_X3_SEG_BLOCKS=1 timeout 200 python tools/bench_quant.py
