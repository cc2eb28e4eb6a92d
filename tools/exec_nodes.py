#!/usr/bin/env python3
"""Lanes of the step executor for the benchmark step: python tools/exec_nodes.py [bench args]"""
import re
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    a = bench.parse()
    a.executor = True
    h = bench.Harness(a, 0, torch.device('cuda', 0))
    for _ in range(6):
        h.step()
    h.settle()
    torch.cuda.synchronize()
    ex = h.captured.executor
    ex._info()
    print(f'{ex.kernels} kernels, lanes {ex.lane_kernels}, {ex.events} events, {ex.waits} waits, '
          f'plan {ex.plan()}')
    tot = [0.0] * ex.lanes
    for i, (lane, us, nw, name) in enumerate(ex.nodes()):
        name = re.sub(r'\(anonymous namespace\)::', '', name).replace('void ', '').split('(')[0][:60]
        tot[lane] += us
        print(f'{i:4d} lane {lane} {us:8.1f} us {"wait" if nw else "    "} {name}')
    print('lane time (calibration, serial):', [round(t, 1) for t in tot])


if __name__ == '__main__':
    main()
