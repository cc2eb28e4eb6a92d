#!/bin/bash
# f32 patch-resident weight gradient: kernel and fold times per decoder stage
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp DVSOF_WGRAD_STREAM=0
OUT=$PWD/gpurun_out/r3wf; mkdir -p $OUT
for cfg in ${CFGS:-"0 512" "32 512" "64 512" "0 256"}; do
set -- ${cfg/_/ }
export DVSOF_WGRAD_PATCH_CT=$1 DVSOF_WGRAD_PATCH_F32_WGS=$2
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 tools/conv_bench.py --reps 3 > $OUT/t.log 2>&1 || exit 1
python3 - "$cfg" $(find $OUT/t -name "*kernel_trace.csv") <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pk = [r for r in rows if 'wgrad_patch' in r['Kernel_Name']]
fold = [r for r in rows if 'subpixel_fold' in r['Kernel_Name']]
for name, ks in (('patch', pk[-4:]), ('fold', fold[-4:])):
    print('CT/WGS', sys.argv[1], name, [(int(r['Grid_Size_X']) // 256, r['Grid_Size_Y'], r['Grid_Size_Z'], round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1)) for r in ks])
PY
rm -rf $OUT/t
done
