#!/usr/bin/env python3
"""One training step as a timeline, from a rocprofv3 kernel trace:
python tools/timeline.py <dir>/x_kernel_trace.csv [step_index] [--brief]
Steps are cut at the optimizer kernel; per queue: busy time and the gaps."""
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:52]


def main():
    path = sys.argv[1]
    args = [a for a in sys.argv[2:] if not a.startswith('--')]
    brief = '--brief' in sys.argv
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    # a captured step begins with set_dynamic_kernel (the optimizer's per-step table); without
    # it steps are cut at the optimizer kernel (one per step unless the update runs per bucket)
    sd = [i for i, r in enumerate(rows) if 'set_dynamic_kernel' in r['Kernel_Name']]
    if len(sd) >= 3:
        k = int(args[0]) if args else max(0, len(sd) - 3)     # a late step: past the executor's plan trials
        a, b = sd[k] - 1, sd[k + 1] - 1
    else:
        ad = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name'] or 'radam' in r['Kernel_Name']]
        k = int(args[0]) if args else len(ad) // 2
        a, b = ad[k], ad[k + 1]
    t0 = rows[a]['e']
    step = rows[a + 1:b + 1]
    print(f'step {k}: {(rows[b]["e"] - t0) / 1e3:.1f} us, {len(step)} kernels')
    copies = []
    mc = path.replace('kernel_trace', 'memory_copy_trace')
    try:
        for r in csv.DictReader(open(mc)):
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            if t0 <= s <= rows[b]['e']:
                copies.append((s, e, r.get('Direction', '')))
    except OSError:
        pass
    queues = sorted({r['Queue_Id'] for r in step})
    for q in queues:
        ks = [r for r in step if r['Queue_Id'] == q]
        busy = sum(r['e'] - r['s'] for r in ks) / 1e3
        print(f'queue {q}: {len(ks)} kernels, busy {busy:.1f} us, first {(ks[0]["s"] - t0) / 1e3:.1f} last {(ks[-1]["e"] - t0) / 1e3:.1f}')
    if brief:
        return
    for s, e, d in copies:
        print(f'{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} copy {d}')
    last = {}
    for r in step:
        q = r['Queue_Id']
        gap = (r['s'] - last[q]) / 1e3 if q in last else 0.0
        last[q] = r['e']
        print(f"{(r['s'] - t0) / 1e3:8.1f} {(r['e'] - t0) / 1e3:8.1f} {(r['e'] - r['s']) / 1e3:7.1f} q{q} gap {gap:6.1f} "
              f"{short(r['Kernel_Name'])} g{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")


if __name__ == '__main__':
    main()
