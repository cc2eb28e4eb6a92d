"""``Model`` -- the plugin class the reference's loader instantiates.

Boundary (reference): ``utils.model.init_model`` imports
``<flownet_path.name>.net`` and calls ``Model(device, **kwargs)`` with kwargs
filtered by this signature (utils/model.py:10-47, utils/options.py:332-347).
``forward(events, timestamps, sample_idx, imsize, raw=True,
intermediate=False)`` returns ``(flows[4] coarse->fine, flow_ts[P,2],
flow_sample_idx[P])`` plus ``(features,)`` iff ``intermediate``
(utils/training.py:59-64; witness DummyNet/net.py:42-80).  ``quantize`` is the
voxeliser entry used by scripts/quantize_preprocessed.py:87-91;
``quantization_layer`` / ``predictor`` are the attribute names
train_flownet.py:50-54,79-85 splits parameter groups on.
"""
import torch
from torch import nn

from . import voxel
from .predictor import Predictor


def get_local_idx(shard_idx):
    """Local index of every element inside its shard and the shard sizes
    (reference DummyNet/net.py:5-39; example [0,0,1,1,2,1,2,2,2] ->
    [0,1,0,1,0,2,1,2,3], [2,3,4]).  Device-side, no host sync except the
    shard count."""
    assert shard_idx.dtype == torch.long
    n = shard_idx.numel()
    bs = int(shard_idx.max()) + 1 if n else 0
    order = torch.argsort(shard_idx, stable=True)
    sizes = torch.bincount(shard_idx, minlength=bs)
    starts = torch.cumsum(sizes, 0) - sizes
    ranks = torch.arange(n, device=shard_idx.device) - starts[shard_idx[order]]
    local = torch.empty_like(ranks)
    local[order] = ranks
    return local, sizes


class VoxelGrid(nn.Module):
    """Event representation (docs/VOXEL_SPEC.md): polarity-signed,
    time-bilinear voxel grid with ``depth`` bins.  No parameters."""

    def __init__(self, depth):
        super().__init__()
        self.depth = depth

    def forward(self, events, t0, t1, batch, height, width):
        """events: the int64 wire columns (utils/dataset.py:961-1020) or the
        compact 9 B/event columns (voxel.is_compact)."""
        if voxel.is_compact(events):
            return voxel.voxelize_compact(events, t0, t1, batch, self.depth,
                                          height, width)
        return voxel.voxelize(events, t0, t1, batch, self.depth, height, width)


class Model(nn.Module):
    def __init__(self, device, prefix_length=0, suffix_length=0,
                 max_sequence_length=1, dynamic_sample_length=False,
                 event_representation_depth=9, activation=None,
                 compute_dtype='f32'):
        super().__init__()
        self.prefix_length = prefix_length
        self.suffix_length = suffix_length
        self.max_sequence_length = max_sequence_length
        self.dynamic_sample_length = dynamic_sample_length
        self.event_representation_depth = event_representation_depth
        self.quantization_layer = VoxelGrid(event_representation_depth)
        self.predictor = Predictor(event_representation_depth, activation,
                                   compute_dtype)
        # strict=True reproduces the reference's host-side assertions (one
        # device sync per call); the train loop turns it off after step one.
        self.strict = True
        self.last_frame_indices = None
        self._layout_cache, self._fast = {}, None
        self.to(device)

    # ---- host/device bookkeeping -------------------------------------
    def _select(self, timestamps, sample_idx, batch):
        """Predicted element = element ``prefix_length`` of every sample
        (DummyNet/net.py:70-78).  -> start, stop frame indices [B] (sorted by
        position), per-sample window t0/t1."""
        D = sample_idx.numel()
        dev = sample_idx.device
        local, sizes = None, None
        if D == batch * (2 + self.prefix_length + self.suffix_length) and \
                not self.strict:
            # uniform, sample-major layout: every index vector is a constant
            # of (D, batch): built once, then ONE gather per step
            key = (D, batch, self.prefix_length, str(dev))
            c = self._layout_cache.get(key)
            if c is None:
                T = D // batch
                first = torch.arange(batch, device=dev) * T
                start = first + self.prefix_length
                c = dict(start=start, stop=start + 1,
                         gather=torch.cat([first, first + T - 1, start,
                                           start + 1]),
                         start32=start.to(torch.int32),
                         stop32=(start + 1).to(torch.int32))
                if len(self._layout_cache) >= 8:
                    self._layout_cache.clear()
                self._layout_cache[key] = c
            g = timestamps[c['gather']]
            self._fast = (c, g)
            return c['start'], c['stop'], g[:batch], g[batch:2 * batch]
        self._fast = None
        local, sizes = get_local_idx(sample_idx)
        if self.strict and not self.dynamic_sample_length:
            assert bool((sizes == (2 + self.prefix_length +
                                   self.suffix_length)).all()), \
                'every sample needs 2 + prefix + suffix timestamps'
        pos = torch.arange(D, device=dev)

        def pick(k):
            key = torch.where(local == k, pos, pos + D)
            return torch.sort(key)[0][:batch]
        start, stop = pick(self.prefix_length), pick(self.prefix_length + 1)
        t0 = torch.full((batch,), float('inf'), device=dev).scatter_reduce(
            0, sample_idx, timestamps, 'amin')
        t1 = torch.full((batch,), float('-inf'), device=dev).scatter_reduce(
            0, sample_idx, timestamps, 'amax')
        return start, stop, t0, t1

    @staticmethod
    def _padded(imsize):
        h, w = int(imsize[0]), int(imsize[1])
        return h, w, (h + 15) // 16 * 16, (w + 15) // 16 * 16

    def _voxels(self, events, t0, t1, batch, imsize):
        h, w, hp, wp = self._padded(imsize)
        return self.quantization_layer(events, t0, t1, batch, hp, wp)

    def quantize(self, events, timestamps, sample_idx, imsize,
                 batch_size=None):
        """events -> float32 [B,C,H,W] (reference caller
        scripts/quantize_preprocessed.py:87-91)."""
        batch = batch_size or int(sample_idx[-1]) + 1
        _, _, t0, t1 = self._select(timestamps, sample_idx, batch)
        h, w, _, _ = self._padded(imsize)
        return self._voxels(events, t0, t1, batch, imsize)[:, :, :h, :w]

    def forward(self, events, timestamps, sample_idx, imsize, raw=True,
                intermediate=False, batch_size=None):
        batch = batch_size or int(sample_idx[-1]) + 1
        if raw and torch.is_grad_enabled() and timestamps.is_cuda:
            self.predictor.mark_step_begin(timestamps.device)
        start, stop, t0, t1 = self._select(timestamps, sample_idx, batch)
        h, w, hp, wp = self._padded(imsize)
        if raw:
            grid = self._voxels(events, t0, t1, batch, imsize)
        else:
            grid = events.float()
            assert grid.shape[0] == batch
            if (hp, wp) != (h, w):
                grid = nn.functional.pad(grid, (0, wp - w, 0, hp - h))
        flows = self.predictor(grid)
        if (hp, wp) != (h, w):      # shrink to the requested size
            flows = tuple(f[:, :, :h // 2 ** i, :w // 2 ** i]
                          for f, i in zip(flows, (3, 2, 1, 0)))
        with torch.no_grad():
            if self._fast is not None:
                c, g = self._fast
                flow_ts = g[2 * batch:].view(2, batch).t()
                self.last_frame_indices = (c['start32'], c['stop32'])
            else:
                flow_ts = torch.stack([timestamps[start], timestamps[stop]], 1)
                self.last_frame_indices = (start.to(torch.int32),
                                           stop.to(torch.int32))
            flow_sample_idx = sample_idx[start]
        add_info = (tuple(), ) if intermediate else tuple()
        return (tuple(flows), flow_ts, flow_sample_idx) + add_info
