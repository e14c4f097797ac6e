#!/bin/bash
# GEMM tile sweep of csrc/vit_packed.hip (MVS_X3P_CFG = "GTT,TI,NW,WJ", read once per process): row tiles per block, columns per block,
# wavefronts, wavefronts along the rows
for cfg in 8,128,8,4 8,128,4,2 8,64,4,2 8,128,8,1 7,128,8,1 7,128,4,1 7,64,4,1; do
  echo "== cfg $cfg"
  q=$cfg; case $cfg in *,64,*) q=8,128,8,4;; esac
  MVS_X3P_CFG=$cfg MVS_X3P_CFG_QKV=$q python tools/bench_x3p.py 2>&1 | grep "x3p" | grep -v "layernorm\|attention"
done
