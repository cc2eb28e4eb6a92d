#!/usr/bin/env python3
"""Host-side enqueue time of one training step vs its GPU time (is the step
launch-bound?).  python tools/host_time.py"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0]]
    a = bench.parse()
    h = bench.Harness(a, 0, torch.device('cuda', 0))
    for _ in range(5):
        h.step()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        h.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'eager: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, with drain {1e3 * (t2 - t0) / n:.3f} ms/step')
    # the same step as one hipGraph replay (capture.CapturedTrainStep)
    a.graph = True
    for _ in range(3):
        h.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        h.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'graph: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, with drain {1e3 * (t2 - t0) / n:.3f} ms/step')
    h.suspend_graph()
    # the captured step through the step executor (csrc/exec.hip)
    a.executor = True
    for _ in range(3):
        h.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        h.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ex = h.captured.executor
    print(f'exec:  host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, with drain {1e3 * (t2 - t0) / n:.3f} ms/step '
          f'({ex.kernels} kernels, lanes {ex.lane_kernels}, {ex.events} events, {ex.waits} waits)')
    t0 = time.perf_counter()
    for _ in range(n):
        ex.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f'exec:  dvsof_exec_launch alone {1e3 * (t1 - t0) / n:.3f} ms/call')
    h.suspend_graph()
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        h.step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(28)


if __name__ == '__main__':
    main()
