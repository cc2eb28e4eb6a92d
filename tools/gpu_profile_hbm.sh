#!/bin/bash
# usage (GPU box, repo root): tools/gpu_profile_hbm.sh <stage>
# kernel-trace stats + FETCH_SIZE / WRITE_SIZE PMC passes (separate runs) of tools/hbm_bench.py
S=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${S}_hbm_stats -o s -- python3 $R/tools/hbm_bench.py > $O/${S}_hbm_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${S}_hbm_fetch -o pmc -- python3 $R/tools/hbm_bench.py > $O/${S}_hbm_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${S}_hbm_write -o pmc -- python3 $R/tools/hbm_bench.py > $O/${S}_hbm_write.log 2>&1 || exit 1
cd $R
python3 tools/pmc_hbm.py $O/${S}_hbm_fetch $O/${S}_hbm_write > $O/${S}_hbm_pmc.csv
cat $O/${S}_hbm_pmc.csv
grep -E "loss_|vox_" $O/${S}_hbm_stats/s_kernel_stats.csv | cut -c1-160
