"""Data augmentation of a wire-format batch on the device: horizontal flip,
nearest-neighbour rotation, crop -- what the reference does per sample on the
CPU in its DataLoader workers (utils/dataset.py:753-769, utils/data.py:24-117,
155-220), here as three kernels over the whole batch.

Events removed by the rotation or the crop keep their slot with x = y = -1
(ignored by the voxeliser) instead of being compacted away, so the batch keeps
static shapes and no host sync is needed.
"""
import numpy as np
import torch

from . import _lib


def random_params(batch_size, in_shape, out_shape, max_angle=30, rng=None):
    """Per-sample parameters drawn like the reference: is_flip ~ Bernoulli(1/2)
    (utils/dataset.py:754), angle ~ U[-max_angle, max_angle)
    (utils/data.py:176-178), crop corner ~ randint(in - out), 0 when equal
    (utils/data.py:107-117).  -> is_flip bool[B], angle f64[B], box i64[B,4]."""
    rng = np.random.default_rng() if rng is None else rng
    is_flip = rng.random(batch_size) < 0.5
    angle = rng.random(batch_size) * 2 * max_angle - max_angle
    box = np.zeros((batch_size, 4), np.int64)
    for k in (0, 1):
        d = in_shape[k] - out_shape[k]
        box[:, k] = rng.integers(0, d, batch_size) if d > 0 else 0
    box[:, 2], box[:, 3] = out_shape
    return is_flip, angle, box


def augment_batch(batch, is_flip, angle, box, device='cuda'):
    """batch: wire format ({'events': {'x','y',...,'sample_index'},
    'images' [D,1,H,W] or [D,H,W], 'sample_idx', ...}) with tensors on any
    device.  is_flip bool[B], angle (degrees) float[B], box int[B,4] =
    (y0, x0, h, w), same (h, w) for every sample.  -> new batch dict (device
    tensors): cropped float32 'images' [D,1,h,w], events with mapped x / y,
    'augmentation_params' updated with box / angle / is_flip."""
    dev = torch.device(device)
    is_flip = np.asarray(is_flip, bool).reshape(-1)
    angle = np.asarray(angle, np.float64).reshape(-1)
    box = np.asarray(box, np.int64).reshape(-1, 4)
    B = is_flip.size
    assert angle.size == B and box.shape[0] == B
    h, w = int(box[0, 2]), int(box[0, 3])
    assert (box[:, 2] == h).all() and (box[:, 3] == w).all(), \
        'one crop shape per batch'
    images = batch['images']
    H, W = int(images.shape[-2]), int(images.shape[-1])
    assert (box[:, 0] >= 0).all() and (box[:, 1] >= 0).all() and \
        (box[:, 0] + h <= H).all() and (box[:, 1] + w <= W).all(), \
        'crop box outside the frame'
    D = images.numel() // (H * W)
    src = images.to(dev).reshape(D, H, W).contiguous()
    if src.dtype not in (torch.uint8, torch.float32):
        src = src.float()
    _lib.require_cuda(src)
    rad = angle * np.pi / 180                       # utils/data.py:181-186
    cs = torch.from_numpy(np.stack([np.cos(rad), np.sin(rad)], 1)).to(dev)
    flip_d = torch.from_numpy(is_flip.astype(np.uint8)).to(dev)
    box_d = torch.from_numpy(box.astype(np.int32)).to(dev)
    frame_sample = batch['sample_idx'].to(dev, torch.int32).contiguous()
    assert frame_sample.numel() == D
    lib, st = _lib.lib(), _lib.stream()
    out = torch.empty(D, 1, h, w, dtype=torch.float32, device=dev)
    _lib.check(lib.dvsof_augment_frames(
        src.data_ptr(), int(src.dtype == torch.uint8), D, H, W,
        frame_sample.data_ptr(), flip_d.data_ptr(), cs.data_ptr(),
        box_d.data_ptr(), h, w, out.data_ptr(), st), 'dvsof_augment_frames')
    ev = {k: v.to(dev) for k, v in batch['events'].items()
          if isinstance(v, torch.Tensor)}
    n = ev['x'].numel()
    lut = None
    if bool((angle != 0).any()):
        lut = torch.empty(B, H, W, dtype=torch.int32, device=dev)
        _lib.check(lib.dvsof_augment_lut(cs.data_ptr(), B, H, W,
                                         lut.data_ptr(), st),
                   'dvsof_augment_lut')
    x = ev['x'].to(torch.long).contiguous()
    y = ev['y'].to(torch.long).contiguous()
    s = ev['sample_index'].to(torch.long).contiguous()
    xo, yo = torch.empty_like(x), torch.empty_like(y)
    _lib.check(lib.dvsof_augment_events(
        x.data_ptr(), y.data_ptr(), s.data_ptr(), n, flip_d.data_ptr(),
        _lib.ptr(lut), box_d.data_ptr(), B, H, W, xo.data_ptr(),
        yo.data_ptr(), st), 'dvsof_augment_events')
    ev['x'], ev['y'] = xo, yo
    result = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v)
              for k, v in batch.items() if k not in ('events', 'images')}
    result['events'], result['images'] = ev, out
    aug = dict(batch.get('augmentation_params') or {})
    aug.update(box=torch.from_numpy(box), angle=torch.from_numpy(angle),
               is_flip=torch.from_numpy(is_flip))
    result['augmentation_params'] = aug
    return result
