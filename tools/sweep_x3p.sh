#!/bin/bash
# GEMM tile / wavefront / stage sweep of csrc/vit_packed.hip (MVS_X3P_CFG = "TI,NW,NS", read once per process)
for cfg in 128,4,2 128,8,2 128,8,3 128,4,3 64,4,2; do
  echo "== plain GEMM cfg $cfg (qkv form: same NW,NS at TI=128)"
  q=128,${cfg#*,}
  MVS_X3P_CFG=$cfg MVS_X3P_CFG_QKV=$q python tools/bench_x3p.py 2>&1 | grep "x3p" | grep -v "layernorm\|attention"
done
