#!/bin/bash
# usage (GPU box, repo root): tools/gpu_quick.sh <stage> [pytest -k expr] -- conv tests, exec-mode bench, per-kernel calibration table
S=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -q -x ${2:+-k "$2"} > $O/${S}_pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/${S}_pytest.txt
timeout -k 10 200 python3 tools/exec_nodes.py > $O/${S}_exec_nodes.txt 2>&1; tail -2 $O/${S}_exec_nodes.txt
for mode in "" "--exec"; do
  timeout -k 10 300 python3 bench.py $mode --no-cpu-baseline --no-roofline > $O/${S}_bench${mode}.json 2> $O/${S}_bench${mode}.err || { echo "bench $mode failed"; tail -20 $O/${S}_bench${mode}.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$O/${S}_bench${mode}.json')); print('$mode', d['value'], d['ms_per_step'])"
done
