"""Seeded synthetic inputs in the reference's batch wire format.

The batch dictionary mirrors what the reference's ``collate_wrapper`` emits
(/root/reference/utils/dataset.py:961-1020, pinned by
tests/dataset/test_dataset.py:260-297): int64 event columns, float32
window-relative timestamps, float32 frames ``[D,1,H,W]`` in 0..255.

Everything is drawn from ``numpy.random.default_rng`` (PCG64, a stable stream
across numpy versions) so that the golden generator that runs next to the
reference, the CPU oracle and the GPU path all see identical bytes
(seed = 1234 + rank).
"""
import numpy as np

WINDOW = 0.04  # seconds between two frames, as in utils/loss.py:232


def box_blur5(img):
    """5x5 box blur with edge replication, float32 accumulate in a fixed order."""
    h, w = img.shape[-2:]
    pad = np.pad(img, [(0, 0)] * (img.ndim - 2) + [(2, 2), (2, 2)], mode='edge')
    acc = np.zeros_like(img, dtype=np.float32)
    for dy in range(5):
        for dx in range(5):
            acc += pad[..., dy:dy + h, dx:dx + w]
    return (acc / np.float32(25.0)).astype(np.float32)


def make_frames(rng, num, height, width):
    """``num`` blurred uniform-u8 frames as float32 [num,1,H,W]."""
    raw = rng.integers(0, 256, size=(num, 1, height, width)).astype(np.float32)
    return box_blur5(raw)


def make_events(rng, batch, height, width, events_per_sample, seq_len=1):
    """Event columns for ``batch`` samples of ``seq_len`` elements each."""
    cols = {k: [] for k in ('x', 'y', 'timestamp', 'polarity',
                            'element_index', 'sample_index')}
    for b in range(batch):
        n = events_per_sample
        t = np.sort(rng.random(n, dtype=np.float32) *
                    np.float32(WINDOW * seq_len))
        cols['x'].append(rng.integers(0, width, size=n, dtype=np.int64))
        cols['y'].append(rng.integers(0, height, size=n, dtype=np.int64))
        cols['timestamp'].append(t.astype(np.float32))
        cols['polarity'].append(
            rng.integers(0, 2, size=n, dtype=np.int64) * 2 - 1)
        cols['element_index'].append(
            np.minimum((t / np.float32(WINDOW)).astype(np.int64), seq_len - 1))
        cols['sample_index'].append(np.full(n, b, dtype=np.int64))
    return {k: (np.concatenate(v) if v else np.zeros(0, np.int64))
            for k, v in cols.items()}


def make_batch(seed, batch, height, width, events_per_sample=None, seq_len=1):
    """A whole synthetic batch (numpy) in the reference wire format."""
    rng = np.random.default_rng(seed)
    if events_per_sample is None:
        events_per_sample = height * width
    events = make_events(rng, batch, height, width, events_per_sample,
                         seq_len)
    num_ts = seq_len + 1
    timestamps = np.tile(np.arange(num_ts, dtype=np.float32) *
                         np.float32(WINDOW), batch).astype(np.float32)
    sample_idx = np.repeat(np.arange(batch, dtype=np.int64), num_ts)
    images = make_frames(rng, batch * num_ts, height, width)
    return {'events': events, 'timestamps': timestamps,
            'sample_idx': sample_idx, 'images': images,
            'augmentation_params': {}, 'size': batch}


def make_flows(seed, batch, shapes, sigma):
    """Random flow fields (pixels of each scale), coarse to fine."""
    rng = np.random.default_rng(seed)
    return [(rng.standard_normal((batch, 2, h, w)) * sigma).astype(np.float32)
            for h, w in shapes]


def scale_shapes(height, width, num_scales=4):
    """Prediction shapes coarse to fine: imsize // 2**i, i = 3..0
    (/root/reference/DummyNet/net.py:60-61)."""
    return [(height // 2 ** i, width // 2 ** i)
            for i in range(num_scales)][::-1]


def to_torch(batch, device='cpu'):
    import torch

    def conv(v):
        if isinstance(v, dict):
            return {k: conv(x) for k, x in v.items()}
        if isinstance(v, np.ndarray):
            return torch.from_numpy(v).to(device)
        return v
    return {k: conv(v) for k, v in batch.items()}
