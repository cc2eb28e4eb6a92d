#!/usr/bin/env python3
"""Practical floor of a short chain of dependent HBM-bound launches at the
sizes of the voxeliser / loss paths (batch 8, 256x256): the same bytes moved
by plain device copies split over the same number of dependent launches.
(device time with the host running ahead, as tools/hbm_bench.py)"""
import torch


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    for name, mb, launches in (('voxelise B=8 (33.6 MB algorithmic)', 33.6, 3),
                               ('loss fwd+bwd+pyramid B=8 (37.6 MB)', 37.6, 4),
                               ('same bytes, one launch', 33.6, 1)):
        n = int(mb * 1e6 / 8 / launches)            # read n floats + write n floats per launch
        src = [torch.empty(n, device='cuda') for _ in range(launches)]
        dst = [torch.empty(n, device='cuda') for _ in range(launches)]

        def run():
            for s, d in zip(src, dst):
                d.copy_(s)
        us = timeit(run)
        gbs = mb * 1e6 / us / 1e3          # MB / us = TB/s; -> GB/s
        print(f'{name:42} {launches} launch(es): {us:6.1f} us = {gbs:5.0f} GB/s '
              f'({100 * gbs / 8000:4.1f} % of 8 TB/s)')


if __name__ == '__main__':
    main()
