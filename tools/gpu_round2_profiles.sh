#!/bin/bash
# Everything profiles/round2/ is made from (GPU box, repo root).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
tools/collect_profiles.sh r2 || exit 1
tools/gpu_profile_hbm.sh r2 || exit 1
cd $R
python3 tools/hbm_bench.py > $O/r2_hbm_kernels.txt 2>&1
python3 tools/launch_floor.py > $O/r2_launch_floor.txt 2>&1
for cfg in "bf16 8" "bf16s 8" "bf16x3 8" "f32 32" "bf16 32" "bf16s 32"; do
  set -- $cfg
  python3 bench.py --dtype $1 --batch $2 --steps 20 --no-cpu-baseline --no-roofline --no-other-modes > $O/r2_mode_$1_b$2.json 2>/dev/null
  python3 -c "import json; d=json.load(open('$O/r2_mode_$1_b$2.json')); print('$1 B=$2', d['value'], d['ms_per_step'])"
done | tee $O/r2_modes.txt
for m in eager graph exec; do
  python3 bench.py --$m --no-cpu-baseline --no-roofline --no-other-modes > $O/r2_bench_$m.json 2>/dev/null
  python3 -c "import json; d=json.load(open('$O/r2_bench_$m.json')); print('$m', d['value'], d['ms_per_step'], d['config']['launch'])" | tee -a $O/r2_modes.txt
done
python3 tools/exec_nodes.py > $O/r2_exec_nodes.txt 2>&1; tail -2 $O/r2_exec_nodes.txt
python3 tools/host_time.py > $O/r2_host_time.txt 2>&1; tail -5 $O/r2_host_time.txt
