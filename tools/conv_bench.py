#!/usr/bin/env python3
"""Per-launch timing of the conv stack inside a real training step
(HIP events on the launch stream).  Usage: python tools/conv_bench.py [--batch 8]"""
import argparse
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--batch', type=int, default=8)
    p.add_argument('--height', type=int, default=256)
    p.add_argument('--width', type=int, default=256)
    p.add_argument('--bins', type=int, default=5)
    p.add_argument('--reps', type=int, default=3)
    p.add_argument('--dtype', default='f32', choices=('f32', 'bf16', 'bf16x3', 'bf16s'))
    a = p.parse_args()
    a.events, a.pool = None, 1
    dev = torch.device('cuda', 0)
    h = bench.Harness(a, 0, dev)
    from dvs_of_training_framework_amd import conv as C
    lib = C._lib.lib()
    for _ in range(2):
        h.step()
    recs = []
    orig = (C.conv_fwd, C.conv_dgrad, C.conv_wgrad)

    def wrap(fn, kind, name):
        def inner(desc, *args, **kw):
            tile = lib.dvsof_conv2d_tile_id(ctypes.byref(desc), kind)
            wino = lib.dvsof_conv2d_winograd_tile(ctypes.byref(desc), kind)
            tile = 40 + wino if wino else tile      # 42 / 44: Winograd F(2x2) / F(4x4)
            e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
            e0.record()
            out = fn(desc, *args, **kw)
            e1.record()
            ho, wo = C.out_size(desc)
            ctot = sum(desc.src[i].C for i in range(desc.nsrc))
            recs.append((name, desc.B * ho * wo, desc.Cout, ctot * desc.ksize ** 2,
                         desc.stride, desc.upsample, tile, bench.conv_flops(desc, kind), e0, e1))
            return out
        return inner
    C.conv_fwd, C.conv_dgrad, C.conv_wgrad = (wrap(orig[0], 0, 'fwd'), wrap(orig[1], 1, 'dgrad'),
                                              wrap(orig[2], 2, 'wgrad'))
    for _ in range(a.reps):
        h.step()
    torch.cuda.synchronize()
    C.conv_fwd, C.conv_dgrad, C.conv_wgrad = orig
    n = len(recs) // a.reps
    print(f'{"kind":6}{"M":>8}{"N":>6}{"K":>7} s u tile {"us":>9} {"TF/s":>7} {"GF":>7}')
    tot_t = tot_f = 0
    for i in range(n):
        name, M, N, K, s, u, tile, fl = recs[i][:8]
        us = min(recs[i + r * n][8].elapsed_time(recs[i + r * n][9]) for r in range(a.reps)) * 1e3
        tot_t += us
        tot_f += fl
        print(f'{name:6}{M:8d}{N:6d}{K:7d} {s} {u} {tile:4d} {us:9.1f} {fl / us / 1e6:7.1f} {fl / 1e9:7.2f}')
    print(f'total {tot_t / 1e3:.3f} ms, {tot_f / tot_t / 1e6:.1f} TF/s over {tot_f / 1e9:.1f} GF')


if __name__ == '__main__':
    main()
