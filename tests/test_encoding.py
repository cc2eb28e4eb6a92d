"""Encoded / quantized batch formats against the reference tests' literal
goldens (tests/dataset/test_encoding.py:21-268, test_quantization.py:15-136),
bit-exact including dtypes, like the reference's own `compare` helper."""
from pathlib import Path

import numpy as np
import pytest
import torch

from dvs_of_training_framework_amd import encoding as enc

GOLD = torch.load(Path(__file__).parent / 'golden' / 'encoding.pt', weights_only=True)


def compare(computed, groundtruth, prefix=''):
    # semantics of /root/reference/tests/utils.py:77-100 (restated)
    if isinstance(computed, torch.Tensor):
        assert isinstance(groundtruth, torch.Tensor), prefix
        assert computed.dtype == groundtruth.dtype, (prefix, computed.dtype, groundtruth.dtype)
        assert torch.equal(computed, groundtruth), prefix
        return
    if isinstance(computed, int):
        assert isinstance(groundtruth, int) and computed == groundtruth, prefix
        return
    if isinstance(computed, (tuple, list)):
        assert len(computed) == len(groundtruth), prefix
        for i, (a, b) in enumerate(zip(computed, groundtruth)):
            compare(a, b, f'{prefix}.{i}')
        return
    assert isinstance(computed, dict) and isinstance(groundtruth, dict), prefix
    assert set(computed) == set(groundtruth), (prefix, set(computed), set(groundtruth))
    for k in computed:
        compare(computed[k], groundtruth[k], f'{prefix}.{k}')


def test_encode_decode_join():
    g = GOLD['encoding']
    compare(enc.encode_batch(**g['decoded']), g['encoded'])
    compare(enc.decode_batch(g['encoded']), g['decoded'])
    compare(enc.join_batches(g['encoded_parts']), g['encoded'])
    assert enc.join_batches([g['encoded']]) is g['encoded']
    assert enc.join_batches([])['events']['x'].dtype == torch.short


def test_select_encoded_ranges():
    e = GOLD['encoding']['encoded']
    assert len(GOLD['ranges']) == 6
    for case in GOLD['ranges']:
        got = enc.select_encoded_ranges(e['events']['events_per_element'],
                                        e['elements_per_sample'], case['begin'], case['end'])
        compare(got, case['gt'])
    with pytest.raises(AssertionError):
        enc.select_encoded_ranges(e['events']['events_per_element'],
                                  e['elements_per_sample'], 1, 1)


def test_quantized_encode_decode_join():
    g = GOLD['quantized']
    compare(enc.encode_quantized_batch(g['decoded_batch']), g['encoded_batch'])
    compare(enc.decode_quantized_batch(g['encoded_batch']), g['decoded_batch'])
    compare(enc.join_batches(g['encoded_batches']), g['encoded_batch'])
    for part, dec in zip(g['encoded_batches'], g['decoded_batches']):
        compare(enc.decode_quantized_batch(part), dec)
    r = enc.select_quantized_ranges(g['encoded_batch']['channels_per_sample'],
                                    g['encoded_batch']['elements_per_sample'], 1, 3)
    assert r['data'] == {'begin': 2, 'end': 6} and r['timestamps'] == {'begin': 3, 'end': 10}


def test_round_trip_synthetic_batch():
    from dvs_of_training_framework_amd import synthetic
    b = synthetic.to_torch(synthetic.make_batch(3, 4, 32, 48, 300, seq_len=2))
    b['augmentation_params'] = {'idx': torch.arange(4)}
    e = enc.encode_batch(b['events'], b['timestamps'], b['sample_idx'], b['images'],
                         b['augmentation_params'], b['size'])
    d = enc.decode_batch(e)
    for k in b['events']:
        assert torch.equal(d['events'][k], b['events'][k]), k
    assert torch.equal(d['sample_idx'], b['sample_idx']) and d['size'] == 4
    off = enc.sample_event_offsets(e)
    assert off.tolist() == [0, 300, 600, 900, 1200]


@pytest.mark.gpu
def test_voxelize_encoded_matches_wire_format_path():
    """9 B/event columns -> the same grid and bit-exact indices as the int64
    wire format (and the oracle)."""
    from dvs_of_training_framework_amd import synthetic
    from dvs_of_training_framework_amd.voxel import voxelize
    from oracle import cpu_oracle as orc
    B, C, H, W = 4, 5, 96, 128
    for n in (50, 20000):
        b_np = synthetic.make_batch(9, B, H, W, n)
        b = synthetic.to_torch(b_np)
        e = enc.encode_batch(b['events'], b['timestamps'], b['sample_idx'], b['images'], {}, B)
        t0 = torch.zeros(B)
        t1 = torch.full((B,), synthetic.WINDOW)
        got, gbin, glin = enc.voxelize_encoded(e, t0, t1, C, H, W, debug=True)
        want, bin0, lin0 = orc.voxelize(b_np['events'], t0.numpy(), t1.numpy(), B, C, H, W)
        assert np.array_equal(gbin.cpu().numpy(), bin0)
        assert np.array_equal(glin.cpu().numpy(), lin0)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-3, atol=2e-5)
        dev_ev = {k: v.cuda() for k, v in b['events'].items()}
        wire = voxelize(dev_ev, t0.cuda(), t1.cuda(), B, C, H, W)
        np.testing.assert_allclose(got.cpu().numpy(), wire.cpu().numpy(), rtol=1e-3, atol=2e-5)
