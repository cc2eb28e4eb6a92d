#!/usr/bin/env python3
"""python tools/wp_parse.py <stage> <dbg...>: kernel averages of tools/gpu_wino_probe.sh runs"""
import csv
import sys
S = sys.argv[1]
for d in sys.argv[2:]:
    out = []
    for r in csv.DictReader(open(f'gpurun_out/{S}_wp_{d}/p_kernel_stats.csv')):
        if any(k in r['Name'] for k in ('gconv2_kernel', 'wgrad2_kernel', 'wino_')):
            n = r['Name'].split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')
            out.append(f"{n[:44]} x{r['Calls']} {float(r['AverageNs']) / 1e3:.1f}")
    print(f'dbg={d}: ' + ' | '.join(out))
