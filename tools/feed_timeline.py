"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace run of tools/feed_trace.py: the last
steps of the train loop -- per step: span of the kernels, the copies that ran under it."""
import csv
import glob
import sys

d = sys.argv[1]
kf = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
cf = glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True)
kern = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', ''))
        for r in csv.DictReader(open(kf))]
kern.sort()
copies = []
if cf:
    for r in csv.DictReader(open(cf[0])):
        copies.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Direction', ''),
                       r.get('Size', r.get('Bytes', '0'))))
copies.sort()
# steps start at the voxeliser's bucket kernel
starts = [i for i, k in enumerate(kern) if 'vox_bucket' in k[2] or 'vox_compact' in k[2]]
print('kernels', len(kern), 'copies', len(copies), 'steps', len(starts))
for si in range(max(0, len(starts) - 8), len(starts) - 1):
    a, b = starts[si], starts[si + 1]
    t0, t1 = kern[a][0], kern[b][0]
    busy_end = max(k[1] for k in kern[a:b])
    cs = [c for c in copies if c[0] < t1 and c[1] > t0]
    print(f'step {si}: period {(t1 - t0) / 1e3:8.1f} us, kernels {b - a}, last kernel ends at '
          f'{(busy_end - t0) / 1e3:8.1f}, idle before next step {(t1 - busy_end) / 1e3:7.1f}')
    for c in cs:
        print(f'     copy {c[2]:>14} {int(c[3]) / 1e6:8.2f} MB  {(c[0] - t0) / 1e3:8.1f} -> {(c[1] - t0) / 1e3:8.1f} us '
              f'({(c[1] - c[0]) / 1e3:7.1f} us, {int(c[3]) / max(c[1] - c[0], 1):.1f} GB/s)')
# largest gaps between consecutive kernels on the busiest queue in the last step
a, b = starts[-2], starts[-1]
t0 = kern[a][0]
prev_end = None
gaps = []
for k in sorted(kern[a:b]):
    if prev_end is not None and k[0] - prev_end > 3000:
        gaps.append(((k[0] - prev_end) / 1e3, (k[0] - t0) / 1e3, k[2][:50]))
    prev_end = max(prev_end or 0, k[1])
print('gaps > 3 us with NO kernel running in the last step:')
for g in gaps:
    print(f'   {g[0]:7.1f} us before {g[2]} at {g[1]:8.1f}')
