#!/usr/bin/env python3
"""Frame pyramid alone at the large shapes: us per call, written MB, GB/s."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from dvs_of_training_framework_amd import synthetic
from dvs_of_training_framework_amd.loss import Losses
from tools.hbm_bench import timeit

for B, H, W in ((8, 256, 256), (64, 256, 256), (16, 480, 640)):
    shapes = synthetic.scale_shapes(H, W)
    ev = Losses(shapes, B, 'cuda')
    images = torch.rand(2 * B, 1, H, W, device='cuda') * 255
    flows = [torch.zeros(B, 2, h, w, device='cuda') for h, w in shapes]
    us = timeit(lambda: ev._pyramid(flows, images))
    wr = 4 * 2 * B * sum(h * w for h, w in shapes)
    print(f'pyramid B={B} {H}x{W}: {us:.1f} us, writes {wr / 1e6:.1f} MB, {wr / us / 1e3:.0f} GB/s written')
