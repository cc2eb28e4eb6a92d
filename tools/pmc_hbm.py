#!/usr/bin/env python3
"""HBM-side traffic of the voxeliser / loss kernels from two rocprofv3 --pmc
passes of tools/hbm_bench.py (FETCH_SIZE and WRITE_SIZE in SEPARATE passes: the
TCC counters do not fit together).  Units and the gfx950 correction per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are KiB;
FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled.  Dispatches are
grouped by (kernel, grid size): the bench runs each path at several shapes.

  python tools/pmc_hbm.py gpurun_out/<fetch_dir> gpurun_out/<write_dir> > profiles/roundN/x_hbm_pmc.csv
"""
import re
import sys

import pandas as pd

PAT = r'(vox_\w+|voxelize_kernel|count_image_kernel|loss_\w+_kernel|resize_bilinear\w*)'


def load(d, counter):
    c = pd.read_csv(f'{d}/pmc_counter_collection.csv')
    c = c[(c.Counter_Name == counter) & c.Kernel_Name.str.contains(PAT)].copy()
    c['k'] = c.Kernel_Name.map(lambda n: re.search(PAT, n).group(1))
    c['grid'] = c.Grid_Size
    c['dur'] = (c.End_Timestamp - c.Start_Timestamp) / 1e3
    return c.groupby(['k', 'grid']).agg(n=('Counter_Value', 'size'),
                                        val=('Counter_Value', 'mean'), dur=('dur', 'mean'))


def main():
    f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    t = f.join(w, lsuffix='_f', rsuffix='_w', how='outer').fillna(0)
    print('kernel,grid_threads,n,fetch_MB,write_MB,dur_us,hbm_GBps')
    for (k, grid), r in t.iterrows():
        fetch = 2 * r.val_f * 1024 / 1e6           # KiB -> MB, x2 (gfx950)
        write = r.val_w * 1024 / 1e6
        dur = r.dur_f if r.dur_f else r.dur_w
        print(f'{k},{int(grid)},{int(max(r.n_f, r.n_w))},{fetch:.2f},{write:.2f},{dur:.2f},'
              f'{(fetch + write) / dur * 1e3:.1f}')


if __name__ == '__main__':
    main()
