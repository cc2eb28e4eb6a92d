#!/bin/bash
S=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_conv.py -m gpu -q -x -k "bf16 or twins or fixed_batch or conv" > $O/${S}_pytest_bf16s.txt 2>&1
rc=$?; tail -15 $O/${S}_pytest_bf16s.txt
[ $rc -ne 0 ] && exit 1
for dt in bf16 bf16s f32; do
  timeout -k 10 200 python3 bench.py --dtype $dt --no-cpu-baseline --no-roofline > $O/${S}_bench_$dt.json 2> $O/${S}_bench_$dt.err || { echo "$dt bench failed"; tail -5 $O/${S}_bench_$dt.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/${S}_bench_$dt.json')); print('$dt', d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
