#!/bin/bash
# usage (GPU box, repo root): tools/gpu_exec_check.sh <stage> -- step executor: tests, bench eager / exec / graph, host time
S=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_capture.py -m gpu -q -x > $O/${S}_pytest_capture.txt 2>&1
echo "pytest rc=$?"; tail -15 $O/${S}_pytest_capture.txt
for mode in "" "--exec" "--graph"; do
  for dt in f32 bf16; do
    timeout -k 10 300 python3 bench.py $mode --dtype $dt --no-cpu-baseline --no-roofline > $O/${S}_bench${mode}_$dt.json 2> $O/${S}_bench${mode}_$dt.err || { echo "bench $mode $dt failed"; tail -20 $O/${S}_bench${mode}_$dt.err; exit 1; }
    python3 -c "import json,sys; d=json.load(open('$O/${S}_bench${mode}_$dt.json')); print('$mode $dt', d['value'], d['ms_per_step'], d['config'].get('launch',''))"
  done
done
timeout -k 10 200 python3 tools/host_time.py > $O/${S}_host_time.txt 2>&1; head -5 $O/${S}_host_time.txt
timeout -k 10 200 python3 tools/exec_nodes.py > $O/${S}_exec_nodes.txt 2>&1; tail -3 $O/${S}_exec_nodes.txt
