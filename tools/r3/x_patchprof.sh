#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3x; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py -x -q -k "twins" 2>&1 | tail -1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- env DVSOF_WGRAD_STREAM=0 python3 tools/conv_bench.py --dtype bf16s --reps 3 > $OUT/t.log 2>&1
python3 - $(find $OUT/t -name "*kernel_trace.csv") <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pk = [r for r in rows if 'wgrad_patch' in r['Kernel_Name']]
fold = [r for r in rows if 'subpixel_fold' in r['Kernel_Name']]
for name, ks in (('patch', pk[-4:]), ('fold', fold[-4:])):
    print(name, [(r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1)) for r in ks])
PY
rm -rf $OUT/t
python3 bench.py --dtype bf16s --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench bf16s', d['ms_per_step'], d['value'])"
