for cfg in "3 512" "3 1024" "2 512" "2 1024" "1 512" "5 512"; do
  set -- $cfg
  echo "WINO_TILE=$1 K32_BLOCKS=$2"
  DVSOF_WGRAD_STREAM=0 DVSOF_WINO_TILE=$1 DVSOF_GCONV_K32_BLOCKS=$2 python3 tools/conv_bench.py 2>/dev/null | awk '($1=="fwd"||$1=="dgrad") && $4==4608 {printf "%s %s | ", $1, $8} END{print ""}'
done
