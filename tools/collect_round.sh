#!/bin/bash
# Round-4 collection on the GPU box (writes gpurun_out/r4_*; the summaries are copied to profiles/round4/):
#   bench JSON of the driver's command, rocprofv3 kernel stats of it, per-launch conv table, PMC passes
#   (HBM traffic, matrix-pipe utilisation), executor plan, step timeline, HBM-path kernels with their PMC
#   passes, the step inside a loopback exchange (world 2, 50 us) and inside a 1-rank RCCL group
R=${GRAFT_REPO_ROOT:-$PWD}; export GRAFT_REPO_ROOT=$R; O=$R/gpurun_out; mkdir -p $O
cd $R
bash tools/collect_profiles.sh r4 > $O/r4_collect.log 2>&1 || { tail -5 $O/r4_collect.log; exit 1; }
python3 tools/exec_nodes.py > $O/r4_exec_nodes.txt 2>/dev/null || exit 1
bash tools/gpu_profile_hbm.sh r4 > $O/r4_hbm.log 2>&1 || { tail -5 $O/r4_hbm.log; exit 1; }
python3 tools/hbm_bench.py > $O/r4_hbm_kernels.txt 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/r4_t -o t -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-modes --no-roofline --no-train-loop > $O/r4_trace.log 2>&1 || exit 1
cd $R
python3 tools/timeline.py $(find $O/r4_t -name "*kernel_trace.csv") > $O/r4_timeline.txt
DVSOF_LOOPBACK=2:50 python3 bench.py --steps 20 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop > $O/r4_bench_loopback.json 2> $O/r4_bench_loopback.err || { tail -3 $O/r4_bench_loopback.err; exit 1; }
# (with the per-launch roofline leg: it runs with the reducer detached, bench.py)
DVSOF_LOOPBACK=2:50 python3 bench.py --steps 10 --warmup 3 --no-other-modes --no-cpu-baseline --no-train-loop > $O/r4_bench_loopback_roofline.json 2> $O/r4_bench_loopback_roofline.err || { tail -3 $O/r4_bench_loopback_roofline.err; exit 1; }
DVSOF_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop > $O/r4_bench_one_rank_rccl.json 2> $O/r4_bench_one_rank_rccl.err || { tail -3 $O/r4_bench_one_rank_rccl.err; exit 1; }
# keep the summaries, drop the raw traces (the merge back is capped at 64 MiB)
cp $O/r4_stats/*/s_kernel_stats.csv $O/r4_kernel_stats.csv 2>/dev/null || cp $(find $O/r4_stats -name "*kernel_stats.csv" | head -1) $O/r4_kernel_stats.csv
cp $(find $O/r4_hbm_stats -name "*kernel_stats.csv" | head -1) $O/r4_hbm_kernel_stats.csv
rm -rf $O/r4_stats $O/r4_pmc_fetch $O/r4_pmc_write $O/r4_pmc_mfma $O/r4_t $O/r4_hbm_stats $O/r4_hbm_fetch $O/r4_hbm_write
ls -la $O | grep r4_
