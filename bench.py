#!/usr/bin/env python3
"""Headline benchmark: training samples/s of the optical-flow hot path
(BASELINE.json: EV_FlowNet on synthetic 256x256x5-bin event tensors,
batch 8 per GPU, fp32; configs[1] at N=1, weak scaling for N>1).

One step = voxelise the event batch -> predictor forward -> fused multi-scale
loss -> backward (dgrad/wgrad on the matrix cores) -> [RCCL gradient
all-reduce, overlapped] -> fused AdamW-amsgrad step.  Inputs are resident in
HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Both forms work for N > 1: started WITHOUT a launcher (no RANK / WORLD_SIZE
in the environment) the process only parses its arguments and starts its own N
ranks as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
127.0.0.1 rendezvous) BEFORE anything touches the GPU, forwards rank 0's one
JSON line and returns non-zero if any rank fails (`self_launch`).

Rank 0 prints ONE JSON line (contract in the task statement) carrying
`roofline` (dominant conv kernel: FLOPs the matrix cores execute / HIP-event
time on the launch stream / dense peak, so `frac` <= 1; the algorithmic
SURVEY-8d figure beside it; `roofline.hbm`: voxelise and pyramid+warp/loss
call paths against 8 TB/s) and, at N=1, `cpu_baseline` (the CPU port of the
same step timed on the host cores over a bounded sample).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see dvs_of_training_framework_amd/__init__.py
os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "2")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def self_launch(argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start
    the N ranks ourselves.  This process never imports torch and never touches
    the GPU (no exec of a GPU-initialised process either): it spawns N children
    of this same script with the torchrun environment, forwards rank 0's stdout
    (the JSON line) to its own, every other rank's stdout and all stderr to its
    stderr, and waits.  The first rank that exits non-zero ends the run: the
    others are terminated (they would wait for it in a collective) and its exit
    code is returned.  -> exit code, or None when this process is a rank
    itself (or N = 1)."""
    import signal
    import socket
    import subprocess
    import threading
    n = 1
    for i, tok in enumerate(argv):
        if tok == '--gpus' and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif tok.startswith('--gpus='):
            n = int(tok.split('=', 1)[1])
    if n <= 1 or 'RANK' in os.environ or 'WORLD_SIZE' in os.environ:
        return None
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    kids = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        kids.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + list(argv),
                                     env=env, stdout=subprocess.PIPE, stderr=sys.stderr, text=True))

    def pump(r, k):
        for line in k.stdout:
            if r == 0:
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write(f'[rank {r}] {line}')
    pumps = [threading.Thread(target=pump, args=(r, k), daemon=True) for r, k in enumerate(kids)]
    for t in pumps:
        t.start()
    deadline = time.time() + float(os.environ.get('DVSOF_LAUNCH_TIMEOUT', '3000'))
    rc, why = 0, None
    while any(k.poll() is None for k in kids):
        bad = [(r, k.returncode) for r, k in enumerate(kids) if k.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            rc = bad[0][1] if bad else 124
            why = (f'rank {bad[0][0]} exited with code {bad[0][1]}' if bad
                   else 'DVSOF_LAUNCH_TIMEOUT reached')
            for k in kids:          # exactly the processes started above
                if k.poll() is None:
                    k.send_signal(signal.SIGTERM)
            t_end = time.time() + 10
            while any(k.poll() is None for k in kids) and time.time() < t_end:
                time.sleep(0.1)
            for k in kids:
                if k.poll() is None:
                    k.kill()
            break
        time.sleep(0.05)
    for k in kids:
        k.wait()
    for t in pumps:
        t.join(2)
    if not rc:
        rc = next((k.returncode for k in kids if k.returncode), 0)
        if rc:
            why = f'a rank exited with code {rc}'
    if rc:
        print(f'bench.py: {n}-rank run failed: {why}; the other ranks were stopped', file=sys.stderr)
    return rc if rc >= 0 else 128 - rc      # killed by a signal: the shell convention


if __name__ == '__main__':
    _rc = self_launch(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

import torch  # noqa: E402

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_BF16_MATRIX_TFLOPS = 2516.6  # dense bf16 (v_mfma_f32_32x32x16_bf16), --dtype bf16 runs
TILE_SHAPES = {
    'gconv': {1: '<2,2,2,2> 128x128', 2: '<2,2,2,1> 128x64', 3: '<2,2,1,1> 64x64',
              4: '<4,1,2,1> 256x32', 5: '<4,1,1,1> 128x32'},
    'wgrad': {1: '<2,2,2,2> 128x128', 2: '<2,2,2,1> 128x64', 3: '<2,2,1,1> 64x64',
              4: '<2,2,1,2> 64x128', 5: '<1,4,1,1> 32x128'},
}


def kernel_name(family, tile, gen, wino=0, patch=0, dtype='f32', kind=0):
    """Name as rocprofv3 reports it (kernel template + tile shape)."""
    if patch == 2:  # csrc/fwd_min.hip, dgrad_min.hip, wgrad_min.hip: nine products per low-res pixel
        return {0: 'fwd_min_f32_kernel (9-product up2+conv3x3, 32 cout x NR x 16 px blocks)',
                1: 'dgrad_min_f32_kernel (9-product, 64 cin x 8 x 16 px blocks)',
                2: 'wgrad_min_f32_kernel (9-product, 32 cout x 64 cin) + slab_reduce'}[kind]
    if patch:   # csrc/fwd_patch.hip, csrc/wgrad_patch.hip: patch-resident decoder kernels
        k = 'f32' if dtype == 'f32' else 'twins'
        return (f'fwd_patch_{k}_kernel 2x16 px blocks' if family == 'gconv'
                else f'wgrad_patch_{k}_kernel 32 cout x 16 views + subpixel_fold')
    if wino:
        # input / gradient transforms + NG component GEMMs + output transform
        ng = (wino + 2) ** 2
        gemm = 'gconv2_kernel<2,2,1,1,1,4,1,0,1>' if family == 'gconv' else 'wgrad2_kernel<2,2,1,1,0,1>'
        return f'winograd F({wino}x{wino},3x3) {family}: wino_*_kernel + {gemm} 64x64 x{ng}'
    if gen == 3:    # csrc/first.hip: planar voxel input, K = 9 C
        return 'first_fwd_kernel 4x32 px' if family == 'gconv' else 'first_wgrad_kernel + reduce'
    if gen == 0:
        return 'wgrad_flat_kernel (VALU, flat members)'
    # (gconv2 instantiations carry two more template arguments in rocprof
    # output: K slices per stage and ring depth, e.g. <2,2,1,1,2,4>)
    return f"{family}{'2' if gen == 2 else ''}_kernel{TILE_SHAPES[family][tile]}"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=30)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--batch', type=int, default=8, help='samples per GPU')
    p.add_argument('--height', type=int, default=256)
    p.add_argument('--width', type=int, default=256)
    p.add_argument('--bins', type=int, default=5)
    p.add_argument('--events', type=int, default=None,
                   help='events per sample (default H*W, SURVEY 8d)')
    p.add_argument('--pool', type=int, default=2,
                   help='distinct resident batches cycled through')
    p.add_argument('--dtype', choices=('f32', 'bf16', 'bf16x3', 'bf16s'), default='f32',
                   help='conv matrix-core operand type (bf16: f32 storage and '
                        'accumulation, operands rounded in registers; bf16s: bf16 twins of '
                        'activations / gradients / prepared weights streamed through LDS)')
    p.add_argument('--fused-optimizer', nargs='?', const='buckets', default='auto',
                   choices=('auto', 'none', 'coarse', 'buckets'),
                   help='update parameters during the backward (optim.fuse_into_backward) '
                        'instead of in optimizer.step(): "buckets" = one launch per gradient '
                        'bucket as soon as its gradients are final (behind its all-reduce '
                        'under data parallelism), "coarse" = decoder + residual parameters in '
                        'one launch when their last weight gradient is done; auto (as '
                        'train_flownet.py --optimizer-in-backward auto): buckets for f32 on '
                        'one GPU, none for the bf16 modes and under data parallelism')
    p.add_argument('--graph', action='store_true',
                   help='replay the step as ONE hipGraph launch (capture.CapturedTrainStep; '
                        'single GPU, bit-identical results)')
    p.add_argument('--exec', dest='executor', action='store_true',
                   help='replay the captured step through the step executor (csrc/exec.hip: plain '
                        'launches from one C call on the eager schedule\'s two streams); the default '
                        'on one GPU')
    p.add_argument('--eager', action='store_true',
                   help='enqueue every step from Python (torch.distributed issues the gradient '
                        'exchange under data parallelism)')
    p.add_argument('--no-other-modes', action='store_true',
                   help='skip the short bf16x3 / bf16 runs reported beside the fp32 headline')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-train-loop', action='store_true',
                   help='skip the train() runs fed from host memory (train_loop in the JSON line)')
    p.add_argument('--no-roofline', action='store_true')
    p.add_argument('--cpu-samples', type=int, default=8,
                   help='samples per step of the cpu_baseline leg (of --batch)')
    p.add_argument('--launch-only', action='store_true',
                   help='print this rank\'s view of the launch (rank, world size, rendezvous) as '
                        'one JSON line and exit without touching the GPU: the CPU test of the '
                        'self-launcher')
    return p.parse_args()


class Harness:
    def __init__(self, a, rank, device):
        from dvs_of_training_framework_amd import synthetic
        from dvs_of_training_framework_amd.loss import init_losses
        from dvs_of_training_framework_amd.net import Model
        from dvs_of_training_framework_amd.optim import FusedAdamW
        torch.manual_seed(1234)            # identical replicas
        self.a, self.device = a, device
        self.model = Model(device, event_representation_depth=a.bins,
                           compute_dtype=getattr(a, 'dtype', 'f32'))
        self.model.train()
        self.opt = FusedAdamW(self.model.predictor.parameters(), lr=1e-3,
                              weight_decay=1e-4, amsgrad=True)
        fo = getattr(a, 'fused_optimizer', 'auto')
        if fo == 'auto':    # measured: +1.5 % in f32 on one GPU; behind the exchange marks of a
            # (1-rank) group the per-bucket waits cost more than the overlap gives (-6 %)
            alone = int(os.environ.get('WORLD_SIZE', '1')) == 1 and os.environ.get('DVSOF_FORCE_DIST') != '1' \
                and not os.environ.get('DVSOF_LOOPBACK')
            fo = 'buckets' if getattr(a, 'dtype', 'f32') == 'f32' and alone else 'none'
        if fo != 'none':
            flush = (5,)
            if os.environ.get('DVSOF_FLUSH_AT'):    # (experiments: other flush points of 'coarse')
                flush = tuple(int(v) for v in os.environ['DVSOF_FLUSH_AT'].split(','))
            self.opt.fuse_into_backward(self.model.predictor,
                                        flush_at=None if fo == 'buckets' else flush)
        self.fused_optimizer = fo
        self.sched = torch.optim.lr_scheduler.LambdaLR(
            self.opt, lambda s: 2 ** (-s / 100000))
        self.losses = init_losses((a.height, a.width), a.batch, self.model,
                                  device, sequence_length=1)
        self.batches = [synthetic.to_torch(synthetic.make_batch(
            1234 + rank + 1000 * i, a.batch, a.height, a.width, a.events),
            device) for i in range(a.pool)]
        self.reducer = None
        self.i = 0
        self.captured = None
        self.captured_all = {}

    def step(self):
        if getattr(self.a, 'executor', False) or \
                (getattr(self.a, 'graph', False) and self.reducer is None):
            return self.graph_step()
        return self.eager_step()

    def graph_step(self):
        # one captured step per resident batch, bound to its buffers (no staging
        # copy: the inputs are in HBM when the timed region starts, as for the
        # eager step)
        k = self.i % len(self.batches)
        self.i += 1
        if k not in self.captured_all:
            from dvs_of_training_framework_amd.capture import CaptureFailed, CapturedTrainStep
            try:
                # under data parallelism the exchange is part of the captured step:
                # the executor issues the bucket all-reduces between its launches
                self.captured = self.captured_all[k] = CapturedTrainStep(
                    self.model, self.losses, self.opt, [0.5, 1, 1], self.device, self.batches[k],
                    executor=bool(getattr(self.a, 'executor', False)), bind=True,
                    reducer=self.reducer)
            except CaptureFailed as e:  # the launch mode, not the product path: this batch's
                # step is done (eagerly); later ones are enqueued from Python, the JSON line says so
                import traceback
                traceback.print_exc()
                self.launch_fallback = f'capture failed ({e}); eager launches'
                self.suspend_graph()
                self.opt.zero_grad(set_to_none=True)
                self.sched.step()
                return e.loss
            loss = self.captured.first_loss
        else:
            loss, _ = self.captured_all[k]()
        self.sched.step()
        return loss

    def settle(self, limit=40):
        """Setup, not warm-up: step until every resident batch has its captured
        step and every executor has chosen its lane plan (capture, calibration
        and 6 timed trial steps per executor)."""
        for _ in range(limit):
            ex = [c.executor for c in self.captured_all.values()]
            if getattr(self, 'launch_fallback', None) or not (
                    getattr(self.a, 'executor', False) or getattr(self.a, 'graph', False)):
                return
            if len(ex) == len(self.batches) and all(e is None or e.plan()[1] for e in ex):
                return
            self.step()

    def suspend_graph(self):
        """Per-launch instrumentation needs eager launches."""
        for c in self.captured_all.values():
            c.close()
        self.captured, self.captured_all = None, {}
        self.a.graph = self.a.executor = False

    def eager_step(self):
        from dvs_of_training_framework_amd.timer import FakeTimer
        from dvs_of_training_framework_amd.training import process_minibatch
        from dvs_of_training_framework_amd.loss import unit_backward
        batch = self.batches[self.i % len(self.batches)]
        self.i += 1
        loss, terms, tags = process_minibatch(
            self.model, batch, FakeTimer(), self.device, True, self.losses,
            [0.5, 1, 1])
        unit_backward(loss)     # loss.backward() seeded with a cached device 1.0
        self.model.strict = False
        if self.reducer is not None:
            self.reducer.wait()
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        self.sched.step()
        return loss


def conv_flops(desc, kind):
    from dvs_of_training_framework_amd import conv as C
    ho, wo = C.out_size(desc)
    ctot = sum(desc.src[i].C for i in range(desc.nsrc))
    return 2.0 * desc.B * ho * wo * desc.Cout * ctot * desc.ksize ** 2


def executed_flops(desc, kind, patch=0):
    """FLOPs the MFMA kernels actually issue: the sub-pixel decomposition of
    upsample+3x3 runs 16 instead of 36 tap-products, its minimal form (patch ==
    2: csrc/*_min.hip) 9 (the phased stride-2 data gradient runs exactly its 9:
    phase (py,px) has (1+py)(1+px) taps)."""
    import ctypes
    from dvs_of_training_framework_amd import conv as C
    f = conv_flops(desc, kind)
    if desc.upsample and desc.ksize == 3 and desc.pad == 1:
        return f * (9.0 if patch == 2 else 16.0) / 36.0
    m = C._lib.lib().dvsof_conv2d_winograd_tile(ctypes.byref(desc), kind)
    if m:   # (m+2)^2 products per m x m outputs instead of 9 m^2
        return f * (m + 2) ** 2 / (9.0 * m * m)
    return f


PEAK_HBM_TBS = 8.0      # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def hbm_bytes(a):
    """Algorithmic bytes per step of the two HBM-bound call paths."""
    n_ev = a.batch * (a.events or a.height * a.width)
    vox = n_ev * 44 + a.batch * a.bins * a.height * a.width * 4
    px = sum((a.height >> k) * (a.width >> k) for k in range(4))
    frames = 2 * a.batch
    pyramid = 4 * frames * (a.height * a.width + px)
    loss = 40 * a.batch * px
    return vox, pyramid + loss


def measure_roofline(h, step_ms, steps=3, light=False):
    """HIP-event timing, on torch's current stream (= the launch stream), of
    every conv-stack launch (grouped by kernel template) and of the two
    HBM-bound call paths (voxelise; pyramid + warp/loss forward+backward)."""
    from dvs_of_training_framework_amd import conv as C, loss as L, voxel as V
    lib = C._lib.lib()
    records, hbm_rec = [], {'voxelise': [], 'warp+loss': []}
    orig = (C.conv_fwd, C.conv_dgrad, C.conv_wgrad)
    orig_vox, orig_fused = V.voxelize, L.Losses.fused

    def pair():
        return (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def wrap(fn, kind, family):
        def inner(desc, *args, **kw):
            import ctypes
            tile = lib.dvsof_conv2d_tile_id(ctypes.byref(desc), kind)
            gen = lib.dvsof_conv2d_kernel_generation(ctypes.byref(desc), kind)
            wino = lib.dvsof_conv2d_winograd_tile(ctypes.byref(desc), kind)
            e0, e1 = pair()
            e0.record()
            out = fn(desc, *args, **kw)
            e1.record()
            patch = lib.dvsof_conv2d_last_patch(kind)
            records.append((kernel_name(family, tile, gen, wino, patch, h.a.dtype, kind), conv_flops(desc, kind),
                            e0, e1, executed_flops(desc, kind, patch)))
            return out
        return inner

    last_call = {}

    def timed(fn, key):
        def inner(*args, **kw):
            e0, e1 = pair()
            e0.record()
            out = fn(*args, **kw)
            e1.record()
            hbm_rec[key].append((e0, e1))
            last_call[key] = (fn, args, kw)
            return out
        return inner

    def alone(key, reps=20):
        """The same call path with the GPU to itself (the in-situ figure shares
        the GPU with the other lane's kernels): seconds per call."""
        if key not in last_call:
            return 0.0
        fn, args, kw = last_call[key]
        with torch.no_grad():
            for _ in range(3):
                fn(*args, **kw)
            torch.cuda.synchronize()
            e0, e1 = pair()
            # behind a ~10 ms spin kernel: the host runs ahead and the events
            # bracket back-to-back device work (these paths are shorter than
            # their enqueue from Python)
            torch.cuda._sleep(25_000_000)
            e0.record()
            for _ in range(reps):
                fn(*args, **kw)
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    def collect():
        del records[:]
        for v in hbm_rec.values():
            del v[:]
        C.conv_fwd = wrap(orig[0], 0, 'gconv')
        C.conv_dgrad = wrap(orig[1], 1, 'gconv')
        C.conv_wgrad = wrap(orig[2], 2, 'wgrad')
        V.voxelize = timed(orig_vox, 'voxelise')
        L.Losses.fused = timed(orig_fused, 'warp+loss')
        try:
            for _ in range(steps):
                h.step()
            torch.cuda.synchronize()
        finally:
            C.conv_fwd, C.conv_dgrad, C.conv_wgrad = orig
            V.voxelize, L.Losses.fused = orig_vox, orig_fused
        out = {}
        for name, fl, e0, e1, xfl in records:
            d = out.setdefault(name, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += fl
            d[2] += e0.elapsed_time(e1) * 1e-3
            d[3] += xfl
        hbm = {k: sum(e0.elapsed_time(e1) for e0, e1 in v) * 1e-3 / max(len(v), 1)
               for k, v in hbm_rec.items()}
        return out, hbm
    agg, hbm_t = collect()      # as timed: two backward streams, launches overlap
    hbm_alone = {k: alone(k) for k in hbm_rec}
    # the same launches one at a time (second backward stream off): what a kernel
    # does when it has the GPU to itself
    prev = os.environ.get('DVSOF_WGRAD_STREAM')
    os.environ['DVSOF_WGRAD_STREAM'] = '0'
    try:
        alone, _ = collect()
    finally:
        if prev is None:
            del os.environ['DVSOF_WGRAD_STREAM']
        else:
            os.environ['DVSOF_WGRAD_STREAM'] = prev
    peak = PEAK_BF16_MATRIX_TFLOPS if getattr(h.a, 'dtype', 'f32') != 'f32' \
        else PEAK_F32_MATRIX_TFLOPS
    T = 1e12

    EB = 2 if getattr(h.a, 'dtype', 'f32') == 'bf16s' else 4   # bytes per operand element in LDS
    LDS_DMA_TBS = 6.3       # L2 -> LDS stream, chip-wide (MI355X_MICROARCH.md: ldsdma-fill 6.4)

    def lds_stream(name, v):
        """Bytes the general implicit-GEMM kernels move L2 -> LDS per launch: a BM x BN tile
        streams (BM + BN) K elements for 2 BM BN K FLOPs, so bytes = executed FLOPs / 2 x
        (1 / BM + 1 / BN) x element size -- the bound that applies to the bf16 modes (the
        matrix pipes are idle 85-95 % of the time there).  None for the patch-resident /
        nine-product / first-layer / Winograd-transform groups (they stream a patch once)."""
        import re
        m = re.search(r'(?:gconv2|wgrad2)_kernel<[^>]*> (\d+)x(\d+)$', name)
        if not m:
            return None
        bm, bn = int(m.group(1)), int(m.group(2))
        nbytes = v[3] / 2.0 * (1.0 / bm + 1.0 / bn) * EB
        return {'bytes_per_launch': round(nbytes / v[0]), 'TBps': round(nbytes / v[2] / 1e12, 2),
                'frac_of_lds_dma_peak': round(nbytes / v[2] / 1e12 / LDS_DMA_TBS, 3)}

    def rates(v):       # (launches, algorithmic flops, seconds, executed flops)
        return {'launches': v[0] // steps, 'avg_launch_us': round(v[2] / v[0] * 1e6, 2),
                'ms_per_step': round(v[2] / steps * 1e3, 3),
                'executed_tflops': round(v[3] / v[2] / T, 2),
                'frac': round(v[3] / v[2] / T / peak, 4),
                'algorithmic_tflops': round(v[1] / v[2] / T, 2)}
    dom = max(agg, key=lambda k: agg[k][2])
    n, fl, sec, xfl = agg[dom]
    tot = [sum(v[i] for v in agg.values()) for i in range(4)]
    tot_alone = [sum(v[i] for v in alone.values()) for i in range(4)]
    vox_b, loss_b = hbm_bytes(h.a)
    # HBM-bound call paths: algorithmic bytes / device time of the WHOLE
    # call path, back to back with the GPU to itself (enqueued behind a spin kernel so
    # that the host runs ahead: the paths are shorter than their enqueue from Python, so
    # an event pair around one call inside the step would time the host, not the GPU)
    hbm = {}
    for key, nbytes in (('voxelise', vox_b), ('warp+loss', loss_b)):
        ta = hbm_alone.get(key, 0.0)
        if ta > 0:
            hbm[key] = {'bound': 'hbm', 'algorithmic_bytes': nbytes,
                        'call_path_us': round(ta * 1e6, 2),
                        'achieved': round(nbytes / ta / 1e9, 1), 'peak': PEAK_HBM_TBS * 1e3,
                        'unit': 'GB/s', 'frac': round(nbytes / ta / 1e12 / PEAK_HBM_TBS, 4),
                        'shape': f'B={h.a.batch} {h.a.height}x{h.a.width}x{h.a.bins} (the benchmark step)'}
    # ... and where bandwidth can show: at batch 8 both paths are one round of resident
    # workgroups (a chain of dependent memory round trips)
    try:
        if light:       # other_modes: the matrix-core figures only
            raise StopIteration
        from tools import hbm_bench
        large = {}
        for name, fn, args in (
                ('voxelise B=64 256x256x5, 65536 ev/sample', hbm_bench.voxel_case, (64, 5, 256, 256, 65536)),
                ('voxelise B=4 512x512x12, 1M ev/sample', hbm_bench.voxel_case, (4, 12, 512, 512, 1_000_000)),
                ('warp+loss B=64 256x256', hbm_bench.loss_case, (64, 256, 256)),
                ('warp+loss B=16 480x640', hbm_bench.loss_case, (16, 480, 640))):
            us, nbytes = fn(*args)
            large[name] = {'algorithmic_bytes': int(nbytes), 'call_path_us': round(us, 1),
                           'achieved': round(nbytes / us / 1e3, 1), 'unit': 'GB/s',
                           'frac': round(nbytes / us / 1e6 / PEAK_HBM_TBS, 4)}
            torch.cuda.empty_cache()
        hbm['large_shapes'] = large
    except StopIteration:
        pass
    except Exception as e:      # noqa: BLE001 -- a diagnostics table, never the headline
        hbm['large_shapes'] = {'error': f'{type(e).__name__}: {e}'}
    roof = {
        # dominant kernel group: FLOPs the matrix cores EXECUTE per launch (the
        # sub-pixel / phase / Winograd forms issue fewer than the layer's
        # algorithmic 2*Ho*Wo*Cout*Cin*k^2) / HIP-event duration / dense peak
        'bound': 'mfma', 'kernel': dom,
        'achieved': round(xfl / sec / T, 2), 'peak': peak, 'unit': 'TFLOP/s',
        'frac': round(xfl / sec / T / peak, 4),
        # HBM-side bytes come from hardware counters, which a process cannot read about
        # itself: the rocprofv3 --pmc passes of this command are under profiles/ (README there)
        'traffic': None,
        'avg_launch_us': round(sec / n * 1e6, 2),
        'gflop_per_launch_executed': round(xfl / n / 1e9, 3),
        'gflop_per_launch_algorithmic': round(fl / n / 1e9, 3),
        # SURVEY 8d figure for the same launches (what the layer computes as
        # specified; may exceed the peak because fewer FLOPs are issued)
        'algorithmic_tflops': round(fl / sec / T, 2),
        # the same launches with the second backward stream off (GPU to itself)
        'single_stream': rates(alone[dom]),
        # every conv launch of a step: sum of per-launch durations (launches of
        # the two backward streams overlap, so these sums exceed wall time)
        'conv_stack': rates(tot), 'conv_stack_single_stream': rates(tot_alone),
        # whole step: executed conv FLOPs of a step / measured step time (the step
        # also voxelises, evaluates the loss and runs the optimizer)
        'step': {'executed_tflops': round(tot[3] / steps / step_ms / 1e9, 2),
                 'frac': round(tot[3] / steps / step_ms / 1e9 / peak, 4),
                 'algorithmic_tflops': round(tot[1] / steps / step_ms / 1e9, 2)},
        'hbm': hbm,
        'per_kernel': {k: dict(rates(v), lds_stream=lds_stream(k, v)) for k, v in agg.items()}}
    # the bound of the dominant group when it is not the matrix pipe: its L2 -> LDS stream
    roof['lds_stream'] = lds_stream(dom, agg[dom])
    # ... and of the largest general-kernel group (the dominant one may be a patch kernel)
    gen = [k for k in agg if lds_stream(k, agg[k])]
    if gen:
        g = max(gen, key=lambda k: agg[k][2])
        roof['lds_stream_largest_general'] = dict(lds_stream(g, agg[g]), kernel=g,
                                                  ms_per_step=round(agg[g][2] / steps * 1e3, 3))
    return roof


def host_cpu():
    """(model string, physical cores usable by this process) from lscpu and
    the affinity mask (a container may see fewer cores than the socket has)."""
    import subprocess
    model, per, sockets, threads_per = 'unknown', None, 1, 1
    try:
        for line in subprocess.run(['lscpu'], capture_output=True, text=True, timeout=10).stdout.splitlines():
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'Model name':
                model = v
            elif k == 'Core(s) per socket':
                per = int(v)
            elif k == 'Socket(s)':
                sockets = int(v)
            elif k == 'Thread(s) per core':
                threads_per = int(v)
    except Exception:       # noqa: BLE001
        pass
    usable = len(os.sched_getaffinity(0))
    physical = per * sockets if per else max(1, usable // max(threads_per, 1))
    if usable < physical * threads_per:      # a CPU share: hardware threads it may run on
        physical = max(1, usable // max(threads_per, 1))
    return model, physical


def cpu_step_rate(B, bins, H, W, events, threads, budget_s, net='evflownet', train=True):
    """samples/s of the CPU port of one step at (B, bins, H, W) on `threads`
    host threads (torch intra-op threads for the ATen predictor + AdamW,
    OpenMP threads over samples for the C voxeliser / loss), over at most
    ~budget_s seconds of timed steps after one warm-up step.
    net='dummy': BASELINE.json configs[0] -- DummyNet (zero flows, no
    parameters, DummyNet/net.py:59-80), forward + loss only (its output carries
    no gradient: tests/training path)."""
    import numpy as np
    from oracle import cpu_oracle as orc
    from oracle.ref_model import ref_predictor
    from dvs_of_training_framework_amd import synthetic
    from dvs_of_training_framework_amd.predictor import Predictor
    torch.set_num_threads(threads)
    orc.use_openmp(threads)
    batch = synthetic.make_batch(1234, B, H, W, events)
    shapes = synthetic.scale_shapes(H, W)
    if net != 'dummy':
        torch.manual_seed(1234)
        state = {k: v.detach().clone().requires_grad_(True)
                 for k, v in Predictor(bins).state_dict().items()}
        opt = torch.optim.AdamW(list(state.values()), lr=1e-3, weight_decay=1e-4, amsgrad=True)

    def step():
        t0 = np.zeros(B, np.float32)
        t1 = np.full(B, synthetic.WINDOW, np.float32)
        grid, _, _ = orc.voxelize(batch['events'], t0, t1, B, bins, H, W)
        ts = batch['timestamps'].reshape(B, 2)
        if net == 'dummy':
            flows = [np.zeros((B, 2, h, w), np.float32) for h, w in shapes]
            orc.losses(flows, ts, np.arange(B), batch['images'], batch['timestamps'],
                       batch['sample_idx'], with_grad=False)
            return
        flows = ref_predictor(state, torch.from_numpy(grid))
        _, loss, grads = orc.losses([f.detach().numpy() for f in flows], ts, np.arange(B),
                                    batch['images'], batch['timestamps'], batch['sample_idx'],
                                    with_grad=train)
        if train:
            torch.autograd.backward(flows, [torch.from_numpy(g) for g in grads])
            opt.step()
            opt.zero_grad(set_to_none=True)
    try:
        step()                              # warm-up
        per_step, t = [], time.perf_counter()
        while len(per_step) < 2 or (time.perf_counter() - t < budget_s and len(per_step) < 30):
            t1 = time.perf_counter()
            step()
            per_step.append(time.perf_counter() - t1)
        dt, n = time.perf_counter() - t, len(per_step)
    finally:
        orc.use_openmp(None)
    rate = sorted(B / x for x in per_step)
    return {'value': round(B * n / dt, 3), 'unit': 'samples/s', 'threads': threads,
            'steps': n, 'batch': B, 'seconds': round(dt, 1),
            'spread': {'min': round(rate[0], 3), 'median': round(rate[n // 2], 3),
                       'max': round(rate[-1], 3)}}


def cpu_baseline(a):
    """CPU port of the same training step (oracle/: C voxeliser + loss with
    OpenMP over samples, PyTorch-CPU restatement of the predictor, torch
    AdamW), timed on the host cores over a bounded sample of the workload
    (n in {1, all physical cores}, CPU model stated, configs 1 and
    4 beside config 2)."""
    model, physical = host_cpu()
    B = max(1, min(a.cpu_samples, a.batch))
    full = cpu_step_rate(B, a.bins, a.height, a.width, a.events, physical, 9.0)
    one = cpu_step_rate(1, a.bins, a.height, a.width, a.events, 1, 1.0)
    other = {
        'configs[0] DummyNet 64x64x3, batch 4, forward + loss only (no parameters)':
            cpu_step_rate(4, 3, 64, 64, None, physical, 1.0, net='dummy'),
        'configs[3] EV_FlowNet 480x640x9, 307200 events/sample, batch 1 of 4 per GPU':
            cpu_step_rate(1, 9, 480, 640, None, physical, 3.0),
    }
    return {'value': full['value'], 'unit': 'samples/s', 'cores': physical, 'kind': 'port',
            'cpu_model': model, 'physical_cores': physical,
            # this figure moves with the host's load and thread placement: the spread is stated
            'spread': full['spread'],
            'sample': f"{full['steps']} steps of batch {B} (of {a.batch}) at {a.height}x{a.width}x{a.bins}, "
                      f"{full['seconds']} s, {physical} threads (ATen predictor + AdamW; C voxeliser / "
                      'loss with OpenMP over samples)',
            'threads': {'1': one, str(physical): full},
            'other_configs': other}


def train_loop_rates(a, device, steps=420, warm=140, legs=('wire', 'compact', 'reference')):
    """samples/s of training.train() FED FROM HOST MEMORY -- the loop the
    reference runs (utils/training.py:138-167) including its host-to-device leg
    (:45-56), which the headline above leaves out (inputs resident in HBM):

      feeder + captured  feed.DeviceFeeder (copy stream, two device slots, the
                         copy of batch n+1 under step n) + train(capture=True)
                         (one captured step bound to each slot);
                         'wire': the int64 columns of the reference's collate
                         (36 B/event moved: element_index is read by no kernel),
                         'compact': the encoded 9 B/event columns
                         (utils/dataset.py:286-289) voxelised directly
      reference leg      train() on pageable host batches, tensor.to(device) on
                         the compute stream at the top of every step, every
                         kernel enqueued from Python

    Host batches are pinned (DataLoader(pin_memory=True), utils/dataloader.py:
    103-108) and cycle through a pool of 4.  (The loop needs ~100 steps to reach
    its steady rate -- two recordings, their trial replays, the allocator:
    120 / 240 / 480 steps measured 2908 / 3016 / 3077 samples/s on the wire leg.)"""
    from dvs_of_training_framework_amd import encoding, synthetic
    from dvs_of_training_framework_amd.feed import DeviceFeeder
    from dvs_of_training_framework_amd.loss import init_losses
    from dvs_of_training_framework_amd.net import Model
    from dvs_of_training_framework_amd.optim import FusedAdamW
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import train

    def pin(v):
        if isinstance(v, dict):
            return {k: pin(x) for k, x in v.items()}
        return v.pin_memory() if torch.is_tensor(v) else v
    wire = [synthetic.to_torch(synthetic.make_batch(4321 + i, a.batch, a.height, a.width, a.events))
            for i in range(4)]

    def as_compact(b):
        enc = encoding.encode_batch(b['events'], b['timestamps'], b['sample_idx'], b['images'],
                                    {}, a.batch)
        return dict(b, events=encoding.compact_events(enc))
    pools = {'wire': wire, 'compact': [as_compact(b) for b in wire]}

    def run(pool, feeder, capture):
        torch.manual_seed(1234)
        model = Model(device, event_representation_depth=a.bins, compute_dtype=a.dtype)
        opt = FusedAdamW(model.predictor.parameters(), lr=1e-3, weight_decay=1e-4, amsgrad=True)
        if capture and a.dtype == 'f32' and getattr(a, 'fused_optimizer', 'auto') in ('auto', 'buckets'):
            opt.fuse_into_backward(model.predictor)     # train_flownet.py --optimizer-in-backward auto
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 2 ** (-s / 100000))
        ev = init_losses((a.height, a.width), a.batch, model, device, sequence_length=1)
        host = [pin(b) for b in pool] if feeder else pool
        loader = ({k: (dict(v) if isinstance(v, dict) else v) for k, v in host[i % len(host)].items()}
                  for i in range(steps + 8))
        fd = DeviceFeeder(loader, device) if feeder else None
        mark = {}

        def clock(step, samples):
            if step == warm:
                torch.cuda.synchronize()
                mark['t0'] = time.perf_counter()
        train(model, device, loader if fd is None else fd, opt, steps, sched, None, ev, timers=FakeTimer(),
              hooks={'clock': clock}, capture=capture, log_every=10 ** 9,
              max_events_per_batch=10 ** 9)
        torch.cuda.synchronize()
        dt = time.perf_counter() - mark['t0']
        out = {'samples_per_s': round(a.batch * (steps - warm) / dt, 1),
               'ms_per_step': round(dt / (steps - warm) * 1e3, 3)}
        if fd is not None:
            out['h2d_MB_per_step'] = round(fd.bytes_moved / fd.batches / 1e6, 2)
        return out
    fed = {k: run(pools[k], True, True) for k in ('wire', 'compact') if k in legs}
    ref = {'wire': run(pools['wire'], False, False)} if 'reference' in legs else {}
    return {'feeder+captured': fed,
            'reference leg (.to(device) per step, eager launches)': ref,
            'steps': steps - warm, 'batch': a.batch,
            'note': 'training.train() fed from host memory; see bench.py:train_loop_rates'}


def main():
    a = parse()
    if os.environ.get('DVSOF_EAGER') == '1':        # (sweep scripts set modes through the environment)
        a.eager = True
    if os.environ.get('DVSOF_DTYPE'):
        a.dtype = os.environ['DVSOF_DTYPE']
    if not (a.eager or a.graph):
        a.executor = True
    launch_mode = (a.graph, a.executor)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if a.launch_only:
        print(json.dumps({'rank': int(os.environ.get('RANK', '0')), 'world': world,
                          'local_rank': int(os.environ.get('LOCAL_RANK', '0')), 'gpus': a.gpus,
                          'master': f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}),
              flush=True)
        return
    if world != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={world} in the environment')
    local = int(os.environ.get('LOCAL_RANK', '0'))
    have = torch.cuda.device_count()        # (counting devices does not initialise the GPU)
    if local >= have:
        raise SystemExit(f'bench.py: rank {os.environ.get("RANK", "0")} of {world} needs GPU {local}, '
                         f'this machine shows {have}: one process per GPU, no rank can start')
    from dvs_of_training_framework_amd import parallel
    rank, local, world = parallel.init_distributed('cuda')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    h = Harness(a, rank, device)
    if world > 1 or os.environ.get('DVSOF_FORCE_DIST') == '1' or os.environ.get('DVSOF_LOOPBACK'):
        # (DVSOF_LOOPBACK="world:delay_us": the loopback communicator -- a late, non-identity
        # exchange on one GPU: what the plumbing of the exchange costs when collectives take time)
        if os.environ.get('DVSOF_LOOPBACK'):
            parallel.claim_streams(device)
        parallel.broadcast_parameters(h.model)
        # ONE communicator -- the C ABI's own (dvsof_comm_create, made here) -- carries the
        # exchange of replayed and eager steps alike; DVSOF_DIRECT_RCCL=0 with --eager:
        # torch.distributed's process group instead (comparison runs)
        h.reducer = parallel.GradReducer()
        h.model.predictor.reducer = h.reducer

    def barrier():
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
    # setup: the first steps size torch's caching allocator, build the per-shape
    # layer plans / optimizer tables and set kernel attributes; they are not
    # part of the W warm-up steps the caller asked for
    for _ in range(4):
        h.step()
    h.settle()
    if world > 1:
        # every rank must have issued the same number of steps (= the same collectives) before
        # the timed region: settling is deterministic, but a rank whose recording failed leaves
        # it early -- level the step counters at the maximum
        t = torch.tensor([h.i], device=device, dtype=torch.int64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        while h.i < int(t):
            h.step()
    for _ in range(a.warmup):
        h.step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = h.step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    final_loss = float(loss.detach())

    out = None
    if rank == 0:
        gb = a.batch * world
        out = {
            'metric': 'training samples/sec (256x256x5 event voxels)',
            'value': round(gb * a.steps / dt, 2), 'unit': 'samples/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': a.dtype, 'data': 'synthetic',
            'config': {
                'workload': f'EV_FlowNet {a.height}x{a.width}x{a.bins}-bin '
                            f'synthetic events, batch {a.batch} per GPU, '
                            + ('fp32 (BASELINE.json configs[1])' if a.dtype == 'f32' else
                               f'{a.dtype} matrix-core operands, f32 storage/accumulate (configs[2] shape per GPU)')
                            + '; full step: voxelise + '
                            'fwd + multi-scale loss + bwd + AdamW-amsgrad',
                'global_batch': gb, 'events_per_sample':
                    a.events or a.height * a.width,
                'parallelism': f'dp{world}', 'final_loss': round(final_loss, 4)},
        }
        if h.reducer is not None:
            # what the communicator that carried the exchange says it spans (ncclCommCount of
            # the C ABI's communicator; the process group's size on the torch.distributed path)
            info = h.reducer.comm_info()
            out['config']['rccl_ranks'] = info['ranks'] if info else torch.distributed.get_world_size()
            if info and info['loopback']:
                out['config']['rccl_ranks'] = 0
                out['config']['loopback_world'] = info['ranks']
            out['config']['exchange'] = (
                f"{info['calls']} bucket all-reduces ({info['elements'] * 4 / 1e6:.1f} MB) on the C ABI's "
                + ('LOOPBACK communicator (no peers: bucket / world, late)' if info['loopback'] else 'RCCL communicator')
                + ' (dvsof_allreduce_bucket)' if info else
                f'{h.reducer.bytes_reduced / 1e6:.1f} MB through torch.distributed (process group nccl)')
    if rank == 0:
        out['config']['launch'] = getattr(h, 'launch_fallback', None) or \
            'eager: every kernel enqueued from Python'
    if rank == 0 and (a.graph or a.executor) and h.captured is not None:
        ex = h.captured.executor
        out['config']['launch'] = 'one hipGraph replay per step' if ex is None else (
            f'step executor: {ex.kernels} kernels on {ex.lanes} streams {ex.lane_kernels}, '
            f'{ex.events} events / {ex.waits} waits per step, one C call'
            + f', lane plan "{ex.plan()[0]}"'
            + ('' if h.fused_optimizer == 'none' else
               f', AdamW per gradient bucket inside the backward ({h.fused_optimizer})')
            + (f'; {ex.marks} exchange marks (bucket all-reduces issued by the executor on the '
               'exchange stream)' if ex.marks else ''))
    if not a.no_roofline:
        h.suspend_graph()       # per-launch HIP events need eager launches
        # The per-launch measurement needs no exchange: under data parallelism it runs with the
        # reducer detached (replicas drift apart from here on -- the timed region is over), so
        # that no rank waits for another in it and a failure stays that rank's own.
        reducer, h.reducer = h.reducer, None
        h.model.predictor.reducer = None
        try:
            roof = measure_roofline(h, dt / a.steps * 1e3)
        except Exception as e:      # noqa: BLE001 -- the headline above stands without it
            if world == 1:
                raise
            roof = {'error': f'{type(e).__name__}: {e}'}
        h.reducer = reducer
        h.model.predictor.reducer = reducer
        if rank == 0:
            out['roofline'] = roof
    if rank == 0 and world == 1 and a.dtype == 'f32' and not a.no_other_modes:
        # the same step in the reduced-precision matrix modes of configs[2] / [4]
        # (short runs; the headline above is the exact-f32 path)
        import copy
        h.suspend_graph()
        del h
        others = {}
        for dt in ('bf16x3', 'bf16', 'bf16s'):
            a2 = copy.copy(a)
            a2.dtype, (a2.graph, a2.executor) = dt, launch_mode
            h2 = Harness(a2, rank, device)
            for _ in range(7):
                h2.step()
            h2.settle()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                h2.step()
            torch.cuda.synchronize()
            d2 = (time.perf_counter() - t1) / 20
            others[dt] = {'samples_per_s': round(a.batch / d2, 1), 'ms_per_step': round(d2 * 1e3, 3)}
            h2.suspend_graph()
            if not a.no_roofline:   # the same roofline as the headline's, against the bf16 matrix peak
                r2 = measure_roofline(h2, d2 * 1e3, steps=2, light=True)
                others[dt]['roofline'] = {
                    'bound': 'mfma', 'peak': r2['peak'], 'unit': 'TFLOP/s', 'kernel': r2['kernel'],
                    'achieved': r2['achieved'], 'frac': r2['frac'], 'avg_launch_us': r2['avg_launch_us'],
                    'step': r2['step'], 'conv_stack_single_stream': r2['conv_stack_single_stream'],
                    # the bound that applies to these modes: bytes through LDS-DMA against the
                    # ~6.3 TB/s the L2 -> LDS stream reaches chip-wide
                    'lds_stream': r2.get('lds_stream'),
                    'lds_stream_largest_general': r2.get('lds_stream_largest_general')}
            del h2
        others['note'] = ('matrix-core operand modes with f32 accumulation: bf16x3 = hi+lo split '
                          'operands, three products (flows within 4e-6, gradients 9e-5 of the exact '
                          'path); bf16 = f32 storage, operands rounded once; bf16s = bf16 twins of '
                          'activations / gradients / weight forms streamed through LDS; roofline: '
                          'executed matrix FLOPs against the DENSE bf16 peak (2516.6 TF/s) -- these '
                          'modes are bound by bytes through LDS / HBM, not by the matrix pipes '
                          '(DESIGN section 4): lds_stream = L2 -> LDS bytes of the general implicit-GEMM '
                          'kernels / their time / 6.3 TB/s')
        out['other_modes'] = others
    if rank == 0 and world == 1 and not a.no_train_loop:
        if 'h' in dir():
            h.suspend_graph()
            del h
        out['train_loop'] = train_loop_rates(a, device)
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(a)
        print(json.dumps(out), flush=True)
    barrier()
    if torch.distributed.is_initialized():
        try:        # the C ABI's own communicator (made for the executor's exchange marks)
            if 'h' in dir() and h.reducer is not None:
                h.suspend_graph()
                h.reducer.close()
        except Exception as e:      # noqa: BLE001 -- shutting down
            print(f'reducer.close: {e}', file=sys.stderr)
        torch.cuda.synchronize()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
