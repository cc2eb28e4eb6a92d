#!/bin/bash
# usage: tools/wgrad_sweep.sh <batch> -- weight-gradient times with every (tile, K splits) forced
# (DVSOF_WGRAD_TILE / DVSOF_WGRAD_SPLITS); one line per setting: "M N K us | ..."
B=${1:-8}
for t in 0 1 2 3 4 5; do
  for s in 0 1 2 3 4 6 8 12 16 24 32 48; do
    if [ $t = 0 ] && [ $s != 0 ]; then continue; fi
    if [ $t != 0 ] && [ $s = 0 ]; then continue; fi
    echo "T=$t S=$s"
    DVSOF_WGRAD_STREAM=0 DVSOF_WGRAD_TILE=$t DVSOF_WGRAD_SPLITS=$s python3 tools/conv_bench.py --batch $B --reps 3 2>/dev/null \
      | awk '$1=="wgrad"{printf "%s %s %s %s %s | ", $2,$3,$4,$7,$8} END{print ""}'
  done
done
