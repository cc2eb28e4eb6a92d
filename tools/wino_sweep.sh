#!/bin/bash
# Winograd tuning sweeps (residual layers, inside a real step, one stream):
# forward / data-gradient GEMM tile and K depth, weight-gradient form, tile and K splits
for cfg in "3 512" "3 1024" "2 512" "2 0" "1 512" "5 512" "4 512"; do
  set -- $cfg
  echo "WINO_TILE=$1 K32_BLOCKS=$2"
  DVSOF_WGRAD_STREAM=0 DVSOF_WINO_TILE=$1 DVSOF_GCONV_K32_BLOCKS=$2 python3 tools/conv_bench.py 2>/dev/null | awk '($1=="fwd"||$1=="dgrad") && $4==4608 {printf "%s %s | ", $1, $8} END{print ""}'
done
for f in 2 4; do for t in 1 2 3 4 5; do for s in 1 2; do
  echo "WINO_WGRAD_F=$f TILE=$t SPLITS=$s"
  DVSOF_WGRAD_STREAM=0 DVSOF_WINO_WGRAD_F=$f DVSOF_WINO_WGRAD_TILE=$t DVSOF_WINO_WGRAD_SPLITS=$s python3 tools/conv_bench.py 2>/dev/null | awk '$1=="wgrad" && $4==4608 {printf "%s %s | ", $1, $8} END{print ""}'
done; done; done
