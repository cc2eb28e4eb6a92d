#!/usr/bin/env python3
"""HBM-side kernels at the sizes of BASELINE.json: achieved algorithmic GB/s
(SURVEY.md section 8d byte counts) from HIP-event timing on the launch stream."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from dvs_of_training_framework_amd import synthetic  # noqa: E402
from dvs_of_training_framework_amd.loss import Losses  # noqa: E402
from dvs_of_training_framework_amd.voxel import voxelize  # noqa: E402


def timeit(fn, n=20):
    """GPU time per call: the calls are enqueued behind a ~10 ms spin kernel so
    that the host runs ahead and the events bracket back-to-back device work
    (at B=8 these paths are shorter than their host-side enqueue time)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(25_000_000)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def loss_case(B, H, W):
    shapes = synthetic.scale_shapes(H, W)
    batch = synthetic.to_torch(synthetic.make_batch(1, B, H, W, 0), 'cuda')
    flows = [torch.from_numpy(f).cuda().requires_grad_(True)
             for f in synthetic.make_flows(2, B, shapes, 2.0)]
    ev = Losses(shapes, B, 'cuda')
    ts = batch['timestamps'].view(B, 2)
    fsi = torch.arange(B, device='cuda')
    idx = (torch.arange(B, device='cuda', dtype=torch.int32) * 2,
           torch.arange(B, device='cuda', dtype=torch.int32) * 2 + 1)

    def run():
        loss, _ = ev.fused(flows, ts, fsi, batch['images'], batch['timestamps'],
                           batch['sample_idx'], frame_indices=idx)
    us = timeit(run)
    px = B * sum(h * w for h, w in shapes)
    alg = 40 * px + 4 * 2 * B * H * W * (1 + sum(h * w for h, w in shapes) / (H * W))
    return us, alg


def voxel_case(B, C, H, W, n):
    rng = np.random.default_rng(3)
    ev = {k: torch.from_numpy(v).cuda() for k, v in synthetic.make_events(rng, B, H, W, n).items()}
    t0 = torch.zeros(B, device='cuda')
    t1 = torch.full((B,), synthetic.WINDOW, device='cuda')
    us = timeit(lambda: voxelize(ev, t0, t1, B, C, H, W))
    return us, B * n * 44 + B * C * H * W * 4


def main():
    print(f'{"kernel path":44}{"us":>10}{"alg MB":>10}{"GB/s":>9}{"% of 6.3TB/s":>14}{"% of 8TB/s":>12}')
    rows = [('loss fused fwd+bwd + pyramid, B=8 256x256', *loss_case(8, 256, 256)),
            ('loss fused fwd+bwd + pyramid, B=64 256x256', *loss_case(64, 256, 256)),
            ('loss fused fwd+bwd + pyramid, B=16 480x640', *loss_case(16, 480, 640)),
            ('voxelise B=8 256x256x5, 65536 ev/sample', *voxel_case(8, 5, 256, 256, 65536)),
            ('voxelise B=64 256x256x5, 65536 ev/sample', *voxel_case(64, 5, 256, 256, 65536)),
            ('voxelise B=4 512x512x12, 1M ev/sample', *voxel_case(4, 12, 512, 512, 1_000_000))]
    for name, us, alg in rows:
        gbs = alg / us / 1e3
        print(f'{name:44}{us:10.1f}{alg / 1e6:10.1f}{gbs:9.0f}{100 * gbs / 6300:14.1f}{100 * gbs / 8000:12.1f}')


if __name__ == '__main__':
    main()
