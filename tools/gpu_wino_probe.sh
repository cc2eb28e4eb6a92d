#!/bin/bash
# usage (GPU box, repo root): tools/gpu_wino_probe.sh <stage> [B] [kind] -- Winograd GEMM under the DVSOF_GCONV_DBG probes
# (probes live in the probe build: make -C dvs_of_training_framework_amd/csrc probes)
S=${1:-x}; B=${2:-8}; K=${3:-fwd}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dbg in ${DBGS:-0 64 320 65 66 68 72 80 96 192}; do
  DVSOF_PROBE_LIB=1 DVSOF_GCONV_DBG=$dbg timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${S}_wp_$dbg -o p -- python3 $R/tools/wino_probe.py $B $K > $O/${S}_wp_$dbg.log 2>&1 || { echo "dbg $dbg failed"; tail -5 $O/${S}_wp_$dbg.log; exit 1; }
  echo "dbg=$dbg $(grep -E 'gconv2_kernel|wgrad2_kernel' $O/${S}_wp_$dbg/p_kernel_stats.csv | awk -F'","' '{printf "%s calls %s avg_ns %s | ", substr($1,1,60), $2, $4}')"
done
