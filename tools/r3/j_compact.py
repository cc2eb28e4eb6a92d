#!/usr/bin/env python3
"""Why is the compact-column train loop slower than the wire-column one? (bench train_loop, round 3)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from dvs_of_training_framework_amd import encoding, synthetic, voxel
from tools.hbm_bench import timeit

B, H, W, C = 8, 256, 256, 5
b = synthetic.to_torch(synthetic.make_batch(1, B, H, W, None))
enc = encoding.encode_batch(b['events'], b['timestamps'], b['sample_idx'], b['images'], {}, B)
comp = {k: v.cuda() for k, v in encoding.compact_events(enc).items()}
wire = {k: v.cuda() for k, v in b['events'].items()}
t0 = torch.zeros(B, device='cuda'); t1 = torch.full((B,), synthetic.WINDOW, device='cuda')
print('dtypes', {k: (v.dtype, tuple(v.shape)) for k, v in comp.items()})
print('wire    voxelize us', timeit(lambda: voxel.voxelize(wire, t0, t1, B, C, H, W)))
print('compact voxelize us', timeit(lambda: voxel.voxelize_compact(comp, t0, t1, B, C, H, W)))
a = voxel.voxelize(wire, t0, t1, B, C, H, W); c = voxel.voxelize_compact(comp, t0, t1, B, C, H, W)
print('equal', torch.equal(a, c))
# padded to a capacity (as the feeder / captured step does)
cap = 1 << 20
pad = {k: (torch.cat([v, torch.full((cap - v.numel(),), -1 if k in 'xy' else 0, dtype=v.dtype, device='cuda')])
           if k != 'sample_event_offsets' else v) for k, v in comp.items()}
print('compact padded to 1M us', timeit(lambda: voxel.voxelize_compact(pad, t0, t1, B, C, H, W)))
padw = {k: torch.cat([v, torch.full((cap - v.numel(),), -1 if k in 'xy' else 0, dtype=v.dtype, device='cuda')]) for k, v in wire.items()}
print('wire padded to 1M us', timeit(lambda: voxel.voxelize(padw, t0, t1, B, C, H, W)))
