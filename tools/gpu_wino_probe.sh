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
  python3 - $dbg $O/${S}_wp_$dbg/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if 'gconv2_kernel' in r['Name'] or 'wgrad2_kernel' in r['Name']:
        print(f"dbg={sys.argv[1]} {r['Name'][5:48]} calls {r['Calls']} avg_us {float(r['AverageNs']) / 1e3:.1f}")
PY
  rm -rf $O/${S}_wp_$dbg
done
