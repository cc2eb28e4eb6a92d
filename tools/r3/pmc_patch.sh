#!/bin/bash
# matrix-pipe / LDS counters of the patch-resident kernels (separate --pmc passes)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp DVSOF_WGRAD_STREAM=0
R=$PWD; OUT=$R/gpurun_out/r3pmc; mkdir -p $OUT
DT=${DT:-f32}
for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "lds:SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$name -o pmc -- python3 $R/tools/conv_bench.py --dtype $DT --reps 2 > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
  python3 - $(find $OUT/$name -name "*counter_collection.csv") <<'PY'
import sys, pandas as pd
c = pd.read_csv(sys.argv[1])
c = c[c.Kernel_Name.str.contains('patch|gconv2_kernel<4')]
c['dur'] = (c.End_Timestamp - c.Start_Timestamp) / 1e3
p = c.pivot_table(index=['Dispatch_Id', 'Kernel_Name'], columns='Counter_Name', values='Counter_Value', aggfunc='sum')
p['dur'] = c.groupby(['Dispatch_Id', 'Kernel_Name']).dur.first()
p = p.reset_index().sort_values('Dispatch_Id').tail(6)
pd.set_option('display.width', 250); pd.set_option('display.max_columns', 30)
p['Kernel_Name'] = p.Kernel_Name.str.replace(r'\(anonymous namespace\)::', '', regex=True).str.slice(0, 40)
print(p.to_string(index=False))
PY
  rm -rf $OUT/$name
done
