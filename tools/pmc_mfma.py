#!/usr/bin/env python3
"""Matrix-core utilisation per conv kernel template from one rocprofv3 --pmc pass
of tools/conv_bench.py:
  tools/pmc.sh <dir> "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
SQ_VALU_MFMA_BUSY_CYCLES counts pipe cycles summed over the chip's 1024 SIMDs
(64 per v_mfma_f32_32x32x2_f32); GRBM_GUI_ACTIVE comes summed over the 8 XCDs;
utilisation = busy / (1024 * GRBM_GUI_ACTIVE / 8).
WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY are fractions of SQ_WAVE_CYCLES.

  python tools/pmc_mfma.py gpurun_out/<dir> > profiles/.../x_mfma_pmc.csv
"""
import re
import sys

import pandas as pd


def main():
    c = pd.read_csv(f'{sys.argv[1]}/pmc_counter_collection.csv')
    c = c[c.Kernel_Name.str.contains(r'gconv\d?_kernel|wgrad\d?_kernel|wino_\w+_kernel|\w+_patch_\w+_kernel|\w+_min_f32_kernel')].copy()
    c['k'] = c.Kernel_Name.map(lambda n: re.sub(
        r'\s', '', re.search(r'((?:gconv\d?|wgrad\d?|wino_\w+|\w+_patch_\w+|\w+_min_f32)_kernel(?:<[^>]*>)?)', n).group(1)))
    c['dur'] = (c.End_Timestamp - c.Start_Timestamp) / 1e3
    p = c.pivot_table(index=['Dispatch_Id', 'k'], columns='Counter_Name', values='Counter_Value',
                      aggfunc='sum')
    p['dur'] = c.groupby(['Dispatch_Id', 'k']).dur.first()
    p = p.reset_index().sort_values('Dispatch_Id')
    # the launches of the last step: from its first-layer kernel (the only gconv v1 launch) on
    first = p.index[p.k.str.startswith('gconv_kernel<')]
    p = p.loc[first[-1]:] if len(first) else p.tail(35)
    print('dispatch,k,dur_us,mfma_util,mfma_insts_M,valu_insts_M,wait_any,wait_inst,active_inst')
    for _, r in p.iterrows():
        util = r.SQ_VALU_MFMA_BUSY_CYCLES / (1024.0 * r.GRBM_GUI_ACTIVE / 8)
        wc = r.SQ_WAVE_CYCLES
        print(f'{int(r.Dispatch_Id)},"{r.k}",{r.dur:.1f},{util:.3f},{r.SQ_INSTS_MFMA / 1e6:.2f},'
              f'{r.SQ_INSTS_VALU / 1e6:.2f},{r.SQ_WAIT_ANY / wc:.3f},{r.SQ_WAIT_INST_ANY / wc:.3f},'
              f'{r.SQ_ACTIVE_INST_ANY / wc:.3f}')


if __name__ == '__main__':
    main()
