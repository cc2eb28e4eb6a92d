#!/usr/bin/env python3
"""Time one conv geometry while sweeping the input channel count: the
intercept of time(K) is the fixed cost of a launch (row setup, ring prologue,
epilogue), the slope the cost per K step.  HIP events on the launch stream.

  python tools/conv_sweep.py fwd|dgrad B H W Cout stride [upsample]
  python tools/conv_sweep.py wgrad  _ H W Cout stride upsample Cin     (sweeps the batch)
"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from dvs_of_training_framework_amd import conv as C  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    kind = sys.argv[1]
    B, H, W, Cout, stride = map(int, sys.argv[2:7])
    up = len(sys.argv) > 7 and sys.argv[7] == '1'
    dev = torch.device('cuda', 0)
    lib = C._lib.lib()
    print(f'{kind} B={B} {H}x{W} Cout={Cout} stride={stride} up={int(up)}')
    print(f'{"Cin":>5}{"K":>7}{"tile":>5}{"gen":>4}{"us":>9}{"exec TF/s":>11}')
    if kind == 'wgrad':     # sweep the K of the weight-gradient GEMM (= pixels) through the batch
        cin = int(sys.argv[8]) if len(sys.argv) > 8 else Cout
        print(f'{"B":>5}{"pixels":>9}{"tile":>5}{"us":>9}{"exec TF/s":>11}')
        for b in (1, 2, 4, 8, 16, 32, 64):
            x = torch.randn(b, H, W, cin, device=dev)
            d = C.make_desc([(x, cin, C.NHWC)], b, H, W, Cout, 3, stride, 1, up, C.ACT_RELU)
            d._keepalive = (x,)
            ho, wo = C.out_size(d)
            g = torch.randn(b, ho, wo, Cout, device=dev)
            dw = torch.empty(Cout, 3, 3, cin, device=dev)
            db = torch.empty(Cout, device=dev)
            us = timeit(lambda: C.conv_wgrad(d, g, dw, db))
            tile = lib.dvsof_conv2d_tile_id(ctypes.byref(d), 2)
            taps = 4 if up else 9
            fl = 2.0 * b * ho * wo * Cout * cin * taps
            print(f'{b:5d}{b * ho * wo:9d}{tile:5d}{us:9.1f}{fl / us / 1e6:11.1f}')
        return
    for cin in (16, 32, 64, 128, 256, 512):
        x = torch.randn(B, H, W, cin, device=dev)
        w = torch.randn(Cout, 3, 3, cin, device=dev) * 0.05
        bias = torch.zeros(Cout, device=dev)
        d = C.make_desc([(x, cin, C.NHWC)], B, H, W, Cout, 3, stride, 1, up, C.ACT_RELU)
        d._keepalive = (x, w)
        ho, wo = C.out_size(d)
        w_fwd, w_dg = C.prepare(d, w, True)
        if kind == 'fwd':
            us = timeit(lambda: C.conv_fwd(d, w_fwd, bias, dev))
            k = 0
        else:
            g = torch.randn(B, ho, wo, Cout, device=dev)
            gx = torch.empty(B, H, W, cin, device=dev)
            us = timeit(lambda: C.conv_dgrad(d, w_dg, g, [dict(p=gx, actsrc=x)], C.ACT_RELU))
            k = 1
        tile = lib.dvsof_conv2d_tile_id(ctypes.byref(d), k)
        gen = lib.dvsof_conv2d_kernel_generation(ctypes.byref(d), k)
        taps = 9
        if up:
            taps = 4
        elif stride == 2 and kind == 'dgrad':
            taps = 16
        fl = 2.0 * B * ho * wo * Cout * cin * taps
        print(f'{cin:5d}{cin * 9:7d}{tile:5d}{gen:4d}{us:9.1f}{fl / us / 1e6:11.1f}')


if __name__ == '__main__':
    main()
