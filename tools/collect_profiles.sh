#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <stage>
# Writes gpurun_out/<stage>_*: bench JSON, rocprofv3 kernel stats of the same command,
# per-launch serial conv table, PMC passes (traffic, matrix-pipe utilisation).
set -e
S=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python3 $R/bench.py > $O/${S}_bench.json 2> $O/${S}_bench.err
DVSOF_WGRAD_STREAM=0 python3 $R/tools/conv_bench.py > $O/${S}_conv_per_launch_serial.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${S}_stats -o s -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes > $O/${S}_stats.log 2>&1
cd $R
tools/pmc.sh ${S}_pmc_fetch FETCH_SIZE
tools/pmc.sh ${S}_pmc_write WRITE_SIZE
tools/pmc.sh ${S}_pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
python3 tools/pmc_traffic.py $O/${S}_pmc_fetch $O/${S}_pmc_write > $O/${S}_traffic_pmc.csv
python3 tools/pmc_mfma.py $O/${S}_pmc_mfma > $O/${S}_mfma_pmc.csv
ls $O | grep "^${S}_"
