#!/usr/bin/env python3
"""Pivot rocprofv3 --pmc csv files (gpurun_out/<dirs>) into one row per conv dispatch
of the LAST training step."""
import sys
import pandas as pd

dirs = sys.argv[1:]
frames = []
for d in dirs:
    c = pd.read_csv(f'gpurun_out/{d}/pmc_counter_collection.csv')
    c = c[c.Kernel_Name.str.contains('gconv2?_kernel|wgrad_kernel')]
    c['dur_us'] = (c.End_Timestamp - c.Start_Timestamp) / 1e3
    c['k'] = c.Kernel_Name.str.extract(r'(gconv2?_kernel<[^>]*>|wgrad_kernel<[^>]*>)')[0].str.replace(' ', '')
    p = c.pivot_table(index=['Dispatch_Id', 'k', 'Grid_Size', 'VGPR_Count', 'LDS_Block_Size'],
                      columns='Counter_Name', values='Counter_Value', aggfunc='sum')
    p['dur_us'] = c.groupby(['Dispatch_Id', 'k', 'Grid_Size', 'VGPR_Count', 'LDS_Block_Size']).dur_us.first()
    p = p.reset_index()
    # keep the last step only (35 conv dispatches per step)
    p = p.sort_values('Dispatch_Id').tail(35).reset_index(drop=True)
    frames.append(p)
base = frames[0][['k', 'Grid_Size', 'VGPR_Count', 'LDS_Block_Size', 'dur_us']].copy()
for p in frames:
    for col in p.columns:
        if col not in ('Dispatch_Id', 'k', 'Grid_Size', 'VGPR_Count', 'LDS_Block_Size', 'dur_us'):
            base[col] = p[col].values
pd.set_option('display.width', 250, 'display.max_columns', 40, 'display.max_rows', 100)
b = base
if 'SQ_WAVE_CYCLES' in b:
    b['mfma%'] = 100 * b.SQ_VALU_MFMA_BUSY_CYCLES / (b.SQ_BUSY_CYCLES * 4 / 8 + 1)  # rough
    b['wait_any%'] = 100 * b.SQ_WAIT_ANY / b.SQ_WAVE_CYCLES
    b['wait_inst%'] = 100 * b.SQ_WAIT_INST_ANY / b.SQ_WAVE_CYCLES
    b['active%'] = 100 * b.SQ_ACTIVE_INST_ANY / b.SQ_WAVE_CYCLES
if 'TCC_HIT_sum' in b:
    b['l2hit%'] = 100 * b.TCC_HIT_sum / (b.TCC_HIT_sum + b.TCC_MISS_sum)
    b['ea_GB'] = b.TCC_EA0_RDREQ_sum * 64 / 1e9
    b['l2req_GB'] = b.TCC_REQ_sum * 128 / 1e9
if 'SQ_LDS_BANK_CONFLICT' in b:
    b['ldsconf%'] = 100 * b.SQ_LDS_BANK_CONFLICT / (b.SQ_LDS_IDX_ACTIVE + 1)
print(b.round(1).to_string())
