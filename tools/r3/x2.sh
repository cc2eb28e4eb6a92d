#!/bin/bash
# patch-resident weight gradient: kernel times per decoder stage for a forced channel tile
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3x; mkdir -p $OUT
for ct in ${CTS:-32 64}; do
export DVSOF_WGRAD_PATCH_CT=$ct
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 tools/conv_bench.py --dtype bf16s --reps 3 > $OUT/t.log 2>&1 || exit 1
python3 - $ct $(find $OUT/t -name "*kernel_trace.csv") <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pk = [r for r in rows if 'wgrad_patch' in r['Kernel_Name']]
fold = [r for r in rows if 'subpixel_fold' in r['Kernel_Name']]
for name, ks in (('patch', pk[-4:]), ('fold', fold[-4:])):
    print('CT', sys.argv[1], name, [(int(r['Grid_Size_X']) // 256, r['Grid_Size_Y'], r['Grid_Size_Z'], round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1)) for r in ks])
PY
rm -rf $OUT/t
python3 bench.py --dtype bf16s --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench bf16s', d['ms_per_step'], d['value'])" || exit 1
done
