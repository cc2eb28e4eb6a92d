#!/usr/bin/env python3
"""Latency of OpticalFlow.__call__ (voxelise + predictor forward, batch 1,
256x256, 65536 events) with eager launches and with the captured HIP graph."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from dvs_of_training_framework_amd.of import OpticalFlow  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    n, H, W = 65536, 256, 256
    ev = [(rng.integers(0, W, n), rng.integers(0, H, n), np.sort(rng.random(n) * 0.04),
           rng.integers(0, 2, n) * 2 - 1)]
    modes = [m == 'graph' for m in sys.argv[1:]] or [False, True]
    for graph in modes:
        torch.manual_seed(0)
        of = OpticalFlow((H, W), model=None, graph=graph, event_representation_depth=5)
        for _ in range(3):
            out = of(ev, [0.0], [0.04])
        torch.cuda.synchronize()
        # per-call latency of the device part (inputs already on the device,
        # result awaited): what a caller of __call__ waits for, minus collate/D2H
        evt, ts, sidx = of._collate(ev, [0.0], [0.04])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            with torch.no_grad():
                if graph:
                    of._replay(evt, ts, sidx, 1)
                else:
                    of._net(evt, ts, sidx, (H, W))
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        print(f'graph={graph}: {dt * 1e3:.3f} ms per call, flow {out.shape} mean {float(np.abs(out).mean()):.5f}')


if __name__ == '__main__':
    main()
