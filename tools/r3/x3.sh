#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3x; mkdir -p $OUT
export DVSOF_WGRAD_PATCH_CT=64 DVSOF_WGRAD_PATCH_MAXS=128
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 tools/conv_bench.py --dtype bf16s --reps 3 > $OUT/t.log 2>&1 || exit 1
python3 - $(find $OUT/t -name "*kernel_trace.csv") <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pk = [r for r in rows if 'wgrad_patch' in r['Kernel_Name']]
fold = [r for r in rows if 'subpixel_fold' in r['Kernel_Name']]
fp = [r for r in rows if 'fwd_patch' in r['Kernel_Name']]
for name, ks in (('patch', pk[-4:]), ('fold', fold[-4:]), ('fwd_patch', fp[-2:])):
    print(name, [(int(r['Grid_Size_X']) // 256, r['Grid_Size_Y'], r['Grid_Size_Z'], round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1)) for r in ks])
PY
rm -rf $OUT/t
