"""Offline quantisation (reference
scripts/quantize_preprocessed.py:60-113) and the raw=False training path
(utils/training.py:54-55).  Upstream has no numeric test at this boundary
(tests/dataset/test_quantization.py pins the encoding only, covered by
tests/test_encoding.py): the checks are round-trip identities."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def batches(n, B, H, W):
    from dvs_of_training_framework_amd import synthetic
    return [synthetic.to_torch(synthetic.make_batch(40 + i, B, H, W, 2000))
            for i in range(n)]


def test_quantize_dataset_groups_files_and_round_trips():
    from dvs_of_training_framework_amd.net import Model
    from dvs_of_training_framework_amd.quantize import (iterate_quantized,
                                                        quantize_dataset)
    B, H, W, C = 2, 64, 64, 5
    model = Model('cuda', event_representation_depth=C)
    model.strict = False
    src = batches(5, B, H, W)
    want = []
    for b in batches(5, B, H, W):
        ev = {k: v.cuda() for k, v in b['events'].items()}
        want.append(model.quantize(ev, b['timestamps'].cuda(),
                                   b['sample_idx'].cuda(), (H, W),
                                   batch_size=B).cpu())
    files = {}
    n = quantize_dataset(src, model, 'cuda', lambda j, joined: files.update({j: joined}),
                         mbs=B, samples_per_file=4, written_indices=(1,))
    assert n == 10
    # ceil(4 / 2) = 2 batches per file; index 1 is taken -> files 0, 2, 3
    assert sorted(files) == [0, 2, 3]
    assert [len(files[j]['elements_per_sample']) for j in (0, 2, 3)] == [4, 4, 2]
    got = []
    for j in (0, 2, 3):
        f = files[j]
        assert f['data'].shape[1:] == (H, W) and f['data'].dtype == torch.float32
        assert bool((f['channels_per_sample'] == C).all())
        assert f['images'].dtype == torch.uint8
        for dec in iterate_quantized(f, B):
            assert dec['size'] == B and tuple(dec['data'].shape) == (B, C, H, W)
            assert dec['sample_idx'].tolist() == [0, 0, 1, 1]
            got.append(dec)
    assert len(got) == 5
    for g, w, b in zip(got, want, batches(5, B, H, W)):
        # float atomics: two voxelisations of the same events agree to the last
        # bits of the accumulation order, not bit for bit
        assert torch.allclose(g['data'], w, rtol=0, atol=1e-5)
        assert torch.equal(g['timestamps'], b['timestamps'])
        assert torch.equal(g['images'], b['images'].to(torch.uint8).float())


def test_quantized_batch_trains_like_the_raw_batch():
    """process_minibatch(is_raw=False) on the offline grid == is_raw=True on
    the events (same loss up to the voxeliser's accumulation order)."""
    from dvs_of_training_framework_amd.loss import init_losses
    from dvs_of_training_framework_amd.net import Model
    from dvs_of_training_framework_amd.quantize import (iterate_quantized,
                                                        quantize_dataset)
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch
    B, H, W, C = 2, 64, 64, 5
    torch.manual_seed(3)
    model = Model('cuda', event_representation_depth=C)
    model.train()
    ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
    model.strict = False
    raw = batches(1, B, H, W)[0]
    # frames must survive the uint8 storage of the encoded form
    raw['images'] = raw['images'].round().clamp(0, 255)
    files = {}
    quantize_dataset([{k: (dict(v) if isinstance(v, dict) else v) for k, v in raw.items()}],
                     model, 'cuda', lambda j, f: files.update({j: f}), mbs=B,
                     samples_per_file=B)
    q = next(iterate_quantized(files[0], B))
    la, ta, _ = process_minibatch(model, raw, FakeTimer(), 'cuda', True, ev, [0.5, 1, 1])
    lb, tb, _ = process_minibatch(model, q, FakeTimer(), 'cuda', False, ev, [0.5, 1, 1])
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(la))
    lb.backward()
    g = [p.grad for p in model.predictor.parameters()]
    assert all(x is not None and bool(torch.isfinite(x).all()) for x in g)


def test_training_from_a_preprocessed_dataset(tmp_path):
    """tests/dataset/test_encoding.py:315-360 restated: a collated batch is
    encoded, written as <n>.hdf5, read back by PreprocessedDataloader and
    trained on.  The decoded (wire-format) batches and the compact 9 B/event
    batches give the same loss (to the voxeliser's summation order)."""
    from dvs_of_training_framework_amd import encoding, hdf5io, synthetic
    if not hdf5io.available():
        pytest.skip('libhdf5 not found')
    import train_flownet as tf
    from dvs_of_training_framework_amd.loss import init_losses
    from dvs_of_training_framework_amd.net import Model
    from dvs_of_training_framework_amd.preprocessed import (PreprocessedDataloader,
                                                             write_encoded_batch)
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch, train
    B, H, W, C = 4, 64, 64, 5
    b = synthetic.to_torch(synthetic.make_batch(31, B, H, W, 6000))
    aug = {'idx': torch.arange(B), 'sequence_length': torch.ones(B, dtype=torch.short),
           'collapse_length': torch.ones(B, dtype=torch.short),
           'box': torch.tensor([[0, 0, H, W]] * B), 'angle': torch.zeros(B),
           'is_flip': torch.zeros(B, dtype=torch.bool)}
    write_encoded_batch(tmp_path / '0.hdf5', encoding.encode_batch(
        b['events'], b['timestamps'], b['sample_idx'], b['images'], aug, B))
    torch.manual_seed(1)
    model = Model('cuda', event_representation_depth=C)
    ev = init_losses((H, W), 2, model, 'cuda', sequence_length=1)
    losses = {}
    for compact in (False, True):
        dl = PreprocessedDataloader(tmp_path, 2, is_raw=True, compact=compact)
        with torch.no_grad():
            loss, _, _ = process_minibatch(model, next(dl), FakeTimer(), 'cuda', True, ev,
                                           [0.5, 1, 1])
        losses[compact] = float(loss)
    assert abs(losses[True] - losses[False]) <= 1e-5 * abs(losses[False])
    # two optimizer steps over the cyclic loader (4 samples, batch 2, compact events)
    args = tf.parse_args(['-m', str(tmp_path / 'model'), '--optimizer', 'ADAM', '-bs', '2', '-mbs',
                          '2', '--height', str(H), '--width', str(W), '-ne', '3',
                          '--event-representation-depth', str(C), '-d', 'cuda:0'])
    optimizer, scheduler = tf.construct_train_tools(args, model)
    seen = []

    class Log:
        def add_scalar(self, t, v, x):
            if t == 'General/Train loss':
                seen.append((x, v))
    train(model, 'cuda', PreprocessedDataloader(tmp_path, 2, is_raw=True, compact=True), optimizer,
          3, scheduler, Log(), ev, timers=FakeTimer())
    assert [x for x, _ in seen] == [2, 4, 6] and all(np.isfinite(v) for _, v in seen)
