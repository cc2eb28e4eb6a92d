"""ctypes front-end of the CPU oracle (oracle/dvsof_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by the product package
(dvs_of_training_framework_amd must fail loudly without its HIP library
instead of routing here).

Host-side logic restated here (numpy):
  resolve_frames  -- utils/loss.py:182-206 (exact float equality + sample id)
  losses          -- utils/loss.py:179-214 (CASCADED pyramid, coarse to fine)
  combined        -- utils/training.py:12-24 + utils/common.py:22-23
"""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = None

c_f = ctypes.POINTER(ctypes.c_float)
c_d = ctypes.POINTER(ctypes.c_double)
c_i64 = ctypes.POINTER(ctypes.c_int64)
c_u64 = ctypes.POINTER(ctypes.c_uint64)
c_i32 = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    so = _DIR / 'libdvsof_oracle.so'
    src = _DIR / 'dvsof_oracle.c'
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(['make', '-C', str(_DIR), '-B', 'libdvsof_oracle.so'],
                       check=True, stdout=subprocess.DEVNULL)
    return so


def use_openmp(threads):
    """bench.py's cpu_baseline leg: switch to libdvsof_oracle_omp.so (the same
    source compiled with -fopenmp: loops over images / samples run on
    ``threads`` cores; results do not depend on the thread count).
    ``threads`` None: back to the serial checker library."""
    global _LIB
    _LIB = None
    if threads is None:
        _FLAVOUR[0] = 'libdvsof_oracle.so'
        return
    import os
    os.environ['OMP_NUM_THREADS'] = str(int(threads))
    _FLAVOUR[0] = 'libdvsof_oracle_omp.so'
    so = _DIR / _FLAVOUR[0]
    src = _DIR / 'dvsof_oracle.c'
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(['make', '-C', str(_DIR), '-B', _FLAVOUR[0]], check=True,
                       stdout=subprocess.DEVNULL)
    lib()
    try:        # (OMP_NUM_THREADS is read once per process: set the count explicitly)
        omp = ctypes.CDLL('libgomp.so.1')
        omp.omp_set_num_threads(int(threads))
    except OSError:
        pass


_FLAVOUR = ['libdvsof_oracle.so']


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(str(build() if _FLAVOUR[0] == 'libdvsof_oracle.so'
                               else _DIR / _FLAVOUR[0]))
        _LIB.orc_resize_bilinear_ac.argtypes = [c_f, c_f] + [ctypes.c_int] * 5
        _LIB.orc_loss_scale_fwd.argtypes = [c_f, c_f, c_f, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, c_d,
                                            c_i64]
        _LIB.orc_loss_scale_bwd.argtypes = [c_f, c_f, c_f, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, c_d,
                                            c_f]
        _LIB.orc_count_image.argtypes = [c_i64, c_i64, ctypes.c_int64,
                                         ctypes.c_int, ctypes.c_int, c_u64]
        _LIB.orc_voxelize.argtypes = [c_i64, c_i64, c_f, c_i64, c_i64,
                                      ctypes.c_int64, c_f, c_f, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      c_f, c_i32, c_i64]
        for f in ('orc_resize_bilinear_ac', 'orc_loss_scale_fwd',
                  'orc_loss_scale_bwd', 'orc_count_image', 'orc_voxelize'):
            getattr(_LIB, f).restype = None
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def resize_bilinear_ac(src, hout, wout):
    src = _f32(src)
    lead = src.shape[:-2]
    hin, win = src.shape[-2:]
    n = int(np.prod(lead)) if lead else 1
    dst = np.empty(lead + (hout, wout), np.float32)
    lib().orc_resize_bilinear_ac(_p(src, c_f), _p(dst, c_f), n, hin, win,
                                 hout, wout)
    return dst


def loss_scale_fwd(prev, nxt, flow):
    prev, nxt, flow = _f32(prev), _f32(nxt), _f32(flow)
    N, _, h, w = flow.shape
    terms = np.zeros(3, np.float64)
    cnt = np.zeros(N, np.int64)
    lib().orc_loss_scale_fwd(_p(prev, c_f), _p(nxt, c_f), _p(flow, c_f), N, h,
                             w, _p(terms, c_d), _p(cnt, c_i64))
    return terms, cnt


def loss_scale_bwd(prev, nxt, flow, g):
    prev, nxt, flow = _f32(prev), _f32(nxt), _f32(flow)
    N, _, h, w = flow.shape
    g = np.ascontiguousarray(g, dtype=np.float64)
    grad = np.empty_like(flow)
    lib().orc_loss_scale_bwd(_p(prev, c_f), _p(nxt, c_f), _p(flow, c_f), N, h,
                             w, _p(g, c_d), _p(grad, c_f))
    return grad


def resolve_frames(flow_ts, flow_sample_idx, timestamps, sample_idx):
    """utils/loss.py:182-206: for prediction p the start (stop) frame is the
    unique d with timestamps[d] == flow_ts[p,0] (flow_ts[p,1]) and
    sample_idx[d] == flow_sample_idx[p]."""
    flow_ts = np.asarray(flow_ts, np.float32)
    timestamps = np.asarray(timestamps, np.float32)
    start, stop = [], []
    for p in range(flow_ts.shape[0]):
        same = np.asarray(sample_idx) == np.asarray(flow_sample_idx)[p]
        for col, dst in ((0, start), (1, stop)):
            hits = np.nonzero(same & (timestamps == flow_ts[p, col]))[0]
            assert hits.size == 1, 'exactly one frame per prediction'
            dst.append(int(hits[0]))
    return np.array(start, np.int64), np.array(stop, np.int64)


def pyramid(images, shapes):
    """utils/loss.py:207-210: scale k resamples scale k-1's image."""
    out, cur = [], _f32(images)
    for (h, w) in shapes:
        cur = resize_bilinear_ac(cur, h, w)
        out.append(cur)
    return out


def losses(flows, flow_ts, flow_sample_idx, images, timestamps, sample_idx,
           weights=(0.5, 1.0, 1.0), with_grad=True, seed_scale=1.0):
    """-> terms [3,K] float64, loss float, grads list (d loss / d flow_k)."""
    start, stop = resolve_frames(flow_ts, flow_sample_idx, timestamps,
                                 sample_idx)
    shapes = [tuple(f.shape[-2:]) for f in flows]
    pyr = pyramid(images, shapes)
    K = len(flows)
    terms = np.zeros((3, K))
    grads = []
    for k, (f, im) in enumerate(zip(flows, pyr)):
        prev, nxt = im[start], im[stop]
        terms[:, k], _ = loss_scale_fwd(prev, nxt, f)
        if with_grad:
            g = np.array(weights, np.float64) / K * seed_scale
            grads.append(loss_scale_bwd(prev, nxt, f, g))
    loss = float(sum(w * terms[i].mean() for i, w in enumerate(weights)))
    return terms, loss, grads


def count_image(x, y, H, W):
    x, y = _i64(x), _i64(y)
    out = np.zeros((H, W), np.uint64)
    lib().orc_count_image(_p(x, c_i64), _p(y, c_i64), x.size, H, W,
                          _p(out, c_u64))
    return out


def voxelize(events, t0, t1, B, C, H, W):
    """-> grid f32 [B,C,H,W], bin0 i32 [n], lin0 i64 [n]."""
    x, y = _i64(events['x']), _i64(events['y'])
    t = _f32(events['timestamp'])
    p, s = _i64(events['polarity']), _i64(events['sample_index'])
    t0, t1 = _f32(t0), _f32(t1)
    n = x.size
    out = np.zeros((B, C, H, W), np.float32)
    bin0 = np.zeros(max(n, 1), np.int32)
    lin0 = np.zeros(max(n, 1), np.int64)
    lib().orc_voxelize(_p(x, c_i64), _p(y, c_i64), _p(t, c_f), _p(p, c_i64),
                       _p(s, c_i64), n, _p(t0, c_f), _p(t1, c_f), B, C, H, W,
                       _p(out, c_f), _p(bin0, c_i32), _p(lin0, c_i64))
    return out, bin0[:n], lin0[:n]


# ---------------------------------------------------------------------------
# Data augmentation: horizontal flip, LUT rotation,
# crop -- restated from /root/reference/utils/dataset.py:753-769 (order: flip,
# rotate, crop), utils/data.py:155-220 (RandomRotation) and :24-42 (EventCrop).
# The event side of the rotation goes through the reference's native
# ``transformation.map`` whose source is absent: PARITY UNPINNED where several
# rotated pixels read the same source pixel (events there go to the SMALLEST
# such pixel index -- this file's rule); bijective cases (multiples of 90
# degrees) are pinned by tests/dataset/test_dataset.py:76-170.
# ---------------------------------------------------------------------------
def rotation_sources(angle, H, W):
    """-> src_y, src_x int64[H,W] (source pixel of every rotated pixel) and the
    validity mask (utils/data.py:166-199)."""
    x, y = np.meshgrid(range(W), range(H))
    x, y = x.ravel().astype(float) - W / 2, y.ravel().astype(float) - H / 2
    rad = angle * np.pi / 180
    c, s = np.cos(rad), np.sin(rad)
    x1 = np.rint(c * x + (-s) * y + W / 2).astype(np.int64)
    y1 = np.rint(s * x + c * y + H / 2).astype(np.int64)
    ok = (x1 >= 0) & (x1 < W) & (y1 >= 0) & (y1 < H)
    return y1.reshape(H, W), x1.reshape(H, W), ok.reshape(H, W)


def augment_sample(images, x, y, is_flip, angle, box):
    """images [n,H,W]; x, y int arrays of the sample's events;
    box = (y0, x0, h, w).  -> images [n,h,w] float32, new x, new y with -1 for
    events the rotation or the crop drops (the reference removes them)."""
    images = np.asarray(images)
    n, H, W = images.shape
    x, y = np.asarray(x, np.int64).copy(), np.asarray(y, np.int64).copy()
    if is_flip:                                   # utils/dataset.py:755-758
        images = images[..., ::-1]
        x = W - x - 1
    sy, sx, ok = rotation_sources(angle, H, W)    # utils/data.py:181-205
    rimg = np.zeros_like(images)
    rimg[:, ok] = images[:, sy[ok], sx[ok]]
    lut = np.full(H * W, -1, np.int64)            # source pixel -> rotated pixel
    dst = np.flatnonzero(ok.ravel())
    src = (sy * W + sx).ravel()[dst]
    # smallest rotated index wins: write in decreasing order
    lut[src[::-1]] = dst[::-1]
    o = lut[y * W + x]
    oy, ox = o // W, o % W
    y0, x0, h, w = (int(v) for v in box)          # utils/data.py:24-42
    keep = (o >= 0) & (ox >= x0) & (ox < x0 + w) & (oy >= y0) & (oy < y0 + h)
    nx = np.where(keep, ox - x0, -1)
    ny = np.where(keep, oy - y0, -1)
    return rimg[:, y0:y0 + h, x0:x0 + w].astype(np.float32), nx, ny
