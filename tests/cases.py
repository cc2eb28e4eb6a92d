"""Shared seeded cases: the SAME definitions tools/make_goldens.py ran through
the reference (keep in sync with SYNTH_CASES there)."""
import numpy as np

from dvs_of_training_framework_amd import synthetic

SYNTH_CASES = {
    'cfg1_64': (11, 4, 64, 64, None, 1.0, 1),
    'cfg2_256': (12, 2, 256, 256, None, 3.0, 1),
    'odd_17x23': (13, 3, 17, 23, [(5, 7), (9, 12), (17, 23)], 2.0, 1),
    'seq3_32x48': (14, 2, 32, 48, None, 1.5, 3),
    'big_flow_40x24': (15, 2, 40, 24, [(10, 6), (40, 24)], 30.0, 1),
}


def synth_case(name):
    seed, B, H, W, shapes, sigma, seq = SYNTH_CASES[name]
    batch = synthetic.make_batch(seed, B, H, W, events_per_sample=0,
                                 seq_len=seq)
    shapes = shapes or synthetic.scale_shapes(H, W)
    flows = synthetic.make_flows(seed + 1000, B, shapes, sigma)
    if name == 'odd_17x23':
        for f in flows:
            f[1] = 0
    ts = batch['timestamps'].reshape(B, seq + 1)
    pre = (seq - 1) // 2
    flow_ts = np.ascontiguousarray(ts[:, pre:pre + 2])
    fsi = np.arange(B, dtype=np.int64)
    return dict(shapes=shapes, B=B, flows=flows, flow_ts=flow_ts,
                flow_sample_idx=fsi, images=batch['images'],
                timestamps=batch['timestamps'],
                sample_idx=batch['sample_idx'])


def fixture_case(fx, i, use_pred, H=246, W=340):
    """tests/loss/test_loss.py:25-65 inputs rebuilt from the committed
    fixture data (frames i, i+1 cropped to 246x340)."""
    im = fx['frames'][i:i + 2, :H, :W].astype(np.float32)[:, None]
    ts = np.array([0, fx['stop'][i] - fx['start'][i]], dtype=np.float32)
    flow = np.zeros((1, 2, H, W), np.float32)
    if use_pred:
        flow = np.ascontiguousarray(
            fx['pred_flow'][:H, :W].transpose(2, 0, 1)[None])
    return dict(shapes=[(H, W)], B=1, flows=[flow],
                flow_ts=ts.reshape(1, 2),
                flow_sample_idx=np.zeros(1, np.int64), images=im,
                timestamps=ts, sample_idx=np.zeros(2, np.int64))
