#!/usr/bin/env python3
"""HBM-side traffic per conv kernel template from two rocprofv3 --pmc passes of
tools/conv_bench.py (tools/pmc.sh <dir> FETCH_SIZE, tools/pmc.sh <dir> WRITE_SIZE;
separate passes: the TCC counters do not fit together).  Units per
/opt/skills/guides/MI355X_MICROARCH.md: both counters are KiB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled.

  python tools/pmc_traffic.py gpurun_out/<fetch_dir> gpurun_out/<write_dir> > profiles/.../x_traffic_pmc.csv
"""
import re
import sys

import pandas as pd


def load(d, counter):
    c = pd.read_csv(f'{d}/pmc_counter_collection.csv')
    c = c[(c.Counter_Name == counter) &
          c.Kernel_Name.str.contains(r'gconv\d?_kernel|wgrad\d?_kernel|wino_\w+_kernel|\w+_patch_\w+_kernel|\w+_min_f32_kernel')].copy()
    c['k'] = c.Kernel_Name.map(lambda n: re.sub(
        r'\s', '', re.search(r'((?:gconv\d?|wgrad\d?|wino_\w+|\w+_patch_\w+|\w+_min_f32)_kernel(?:<[^>]*>)?)', n).group(1)))
    c['dur'] = (c.End_Timestamp - c.Start_Timestamp) / 1e3
    return c.groupby('k').agg(n=('Counter_Value', 'size'), val=('Counter_Value', 'mean'),
                              dur=('dur', 'mean'))


def main():
    f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    t = f.join(w, lsuffix='_f', rsuffix='_w', how='outer').fillna(0)
    print('k,n,fetch_MB,dur,write_MB,hbm_GBps')
    for k, r in t.iterrows():
        fetch = 2 * r.val_f * 1024 / 1e6           # KiB -> MB, x2 (gfx950)
        write = r.val_w * 1024 / 1e6
        dur = r.dur_f if r.dur_f else r.dur_w
        print(f'"{k}",{int(max(r.n_f, r.n_w))},{fetch:.2f},{dur:.2f},{write:.2f},'
              f'{(fetch + write) / dur * 1e3:.2f}')


if __name__ == '__main__':
    main()
