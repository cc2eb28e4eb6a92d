#!/bin/bash
# usage: tools/tile_sweep.sh <batch> -- forward/data-gradient conv times with every MFMA tile forced
# (DVSOF_GCONV_TILE), one line per tile: "kind M N K TF/s | ..."
B=${1:-8}
for t in 0 1 2 3 4 5; do
  echo "TILE=$t"
  DVSOF_WGRAD_STREAM=0 DVSOF_GCONV_TILE=$t python3 tools/conv_bench.py --batch $B --reps 3 \
    | awk '$1=="fwd"||$1=="dgrad"{printf "%s %s %s %s %s | ", $1,$2,$3,$4,$9} END{print ""}'
done
