"""HDF5 container + PreprocessedDataloader against the reference tests'
literals (tests/dataset/test_encoding.py:270-313 read/write + loader incl.
wrap-around; tests/dataset/test_quantization.py:145-166), bit-exact including
dtypes.  The HDF5 backend (libhdf5 via ctypes) is additionally checked against
files h5py wrote -- the reference's own fixtures -- where they are reachable
(this container; they do not travel to the GPU box)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from dvs_of_training_framework_amd import encoding as enc, hdf5io
from tests.test_encoding import GOLD, compare

pytestmark = pytest.mark.skipif(not hdf5io.available(), reason='libhdf5 not found')


def test_hdf5_round_trip_of_every_dtype(tmp_path):
    data = {'i8': torch.tensor([-3, 4], dtype=torch.int8),
            'u8': torch.arange(6, dtype=torch.uint8).view(2, 3),
            'i16': torch.tensor([[1, -2], [3, -4]], dtype=torch.short),
            'i32': torch.tensor([7], dtype=torch.int32),
            'i64': torch.tensor([2 ** 40, -5]),
            'f32': torch.randn(4, 1, 2, 3), 'f64': torch.randn(2, dtype=torch.float64),
            'flag': torch.tensor([True, False, True]),
            'empty': torch.zeros(0, 4, dtype=torch.short),
            'nested': {'inner': {'x': torch.arange(5)}}}
    from dvs_of_training_framework_amd.preprocessed import write_encoded_batch
    path = tmp_path / 'a.hdf5'
    write_encoded_batch(path, data)
    with hdf5io.File(path, 'r') as f:
        assert set(f.keys()) == set(data)
        for k, v in data.items():
            if isinstance(v, dict):
                continue
            got = torch.from_numpy(f[k][...])
            assert got.dtype == v.dtype and torch.equal(got, v), k
            assert f[k].shape == tuple(v.shape)
        assert torch.equal(torch.from_numpy(f['nested']['inner']['x'][1:4]), torch.arange(1, 4))
        assert len(f['f32']) == 4 and f['f32'][1:3].shape == (2, 1, 2, 3)
        assert 'nope' not in f
        with pytest.raises(KeyError):
            f['nope']
    h5dump = Path('/opt/conda/bin/h5dump')
    if h5dump.exists():     # an independent reader agrees on names, types and values
        text = subprocess.run([str(h5dump), str(path)], capture_output=True, text=True).stdout
        assert 'H5T_ENUM' in text and '"TRUE"' in text and 'DATASET "i16"' in text
        assert 'GROUP "nested"' in text and 'H5T_STD_I16LE' in text


def test_reads_files_written_by_h5py(fixtures):
    """The reference's test sequence was written by h5py: every dataset read
    through hdf5io equals the h5dump extraction committed as fixtures.npz."""
    seq = Path('/root/reference/tests/data/seq/000001.hdf5')
    if not seq.exists():
        pytest.skip('reference fixtures are only reachable in the build container')
    with hdf5io.File(seq, 'r') as f:
        ev = f['events'][...]
        assert ev.dtype == np.float64 and np.array_equal(ev, fixtures['events_1'])
        assert np.array_equal(f['image1'][...], fixtures['frames'][1])
        assert np.array_equal(f['image2'][...], fixtures['frames'][2])
        assert float(f['start'][...]) == float(fixtures['start'][1])
        assert float(f['stop'][...]) == float(fixtures['stop'][1])
    pred = Path('/root/reference/tests/data/pred/000001.hdf5')
    with hdf5io.File(pred, 'r') as f:
        assert np.array_equal(f['flow'][...], fixtures['pred_flow'])


def test_read_prepared_batch(tmp_path):
    # tests/dataset/test_encoding.py:270-290
    from dvs_of_training_framework_amd.preprocessed import (read_encoded_batch,
                                                             write_encoded_batch)
    g = GOLD['encoding']
    path = tmp_path / 'b.hdf5'
    write_encoded_batch(path, g['encoded'])
    for (b, e), part in zip(((0, 2), (2, 3)), g['encoded_parts']):
        with hdf5io.File(path, 'r') as f:
            eps = torch.from_numpy(f['elements_per_sample'][...])
            epe = torch.from_numpy(f['events']['events_per_element'][...])
            compare(read_encoded_batch(f, epe, eps, b, e), part)


def test_quantized_read_write(tmp_path):
    # tests/dataset/test_quantization.py:145-166
    from dvs_of_training_framework_amd.preprocessed import (read_encoded_quantized_batch,
                                                             write_encoded_batch)
    g = GOLD['quantized']
    path = tmp_path / 'q.hdf5'
    write_encoded_batch(path, g['encoded_batch'])
    assert path.is_file()
    for (b, e), want in (((0, 3), g['encoded_batch']), ((0, 2), g['encoded_batches'][0]),
                         ((2, 3), g['encoded_batches'][1])):
        with hdf5io.File(path, 'r') as f:
            cps = torch.from_numpy(f['channels_per_sample'][...])
            eps = torch.from_numpy(f['elements_per_sample'][...])
            compare(read_encoded_quantized_batch(f, cps, eps, b, e), want)


def _write_parts(dirname, parts):
    from dvs_of_training_framework_amd.preprocessed import write_encoded_batch
    for i, part in enumerate(parts):
        write_encoded_batch(dirname / f'{i}.hdf5', part)


def test_preprocessed_dataloader(tmp_path):
    # tests/dataset/test_encoding.py:292-313, including the wrap-around case
    from dvs_of_training_framework_amd.preprocessed import PreprocessedDataloader
    parts = GOLD['encoding']['encoded_parts']
    _write_parts(tmp_path, parts)
    dl = PreprocessedDataloader(tmp_path, 2, is_raw=True)
    assert len(dl) == 3
    compare(next(dl), enc.decode_batch(parts[0]))
    assert (tmp_path / '0.info').read_text().strip() == 'size: 2'

    dl = PreprocessedDataloader(tmp_path, 1, is_raw=True)
    dl.set_index(2)
    compare(next(dl), enc.decode_batch(parts[1]))

    dl = PreprocessedDataloader(tmp_path, 3, is_raw=True)
    compare(next(dl), enc.decode_batch(enc.join_batches(parts)))

    dl = PreprocessedDataloader(tmp_path, 5, is_raw=True)     # 3 samples: wraps to file 0
    compare(next(dl), enc.decode_batch(enc.join_batches(parts + [parts[0]])))
    # the position carries on: sample 2 (file 1), file 0, file 1, first sample of file 0
    from dvs_of_training_framework_amd.preprocessed import read_encoded_batch
    with hdf5io.File(tmp_path / '0.hdf5', 'r') as f:
        first = read_encoded_batch(f, torch.from_numpy(f['events']['events_per_element'][...]),
                                   torch.from_numpy(f['elements_per_sample'][...]), 0, 1)
    compare(next(dl), enc.decode_batch(enc.join_batches([parts[1], parts[0], parts[1], first])))
    # set_index wraps modulo the dataset length
    dl = PreprocessedDataloader(tmp_path, 1, is_raw=True)
    dl.set_index(3 * 7 + 2)
    compare(next(dl), enc.decode_batch(parts[1]))
    with pytest.raises(AssertionError):
        PreprocessedDataloader(tmp_path / 'nothing_here', 1, is_raw=True)


def test_preprocessed_dataloader_quantized_and_file_order(tmp_path):
    from dvs_of_training_framework_amd.preprocessed import PreprocessedDataloader
    g = GOLD['quantized']
    # numeric, not lexicographic, order: 2.hdf5 before 10.hdf5
    from dvs_of_training_framework_amd.preprocessed import write_encoded_batch
    write_encoded_batch(tmp_path / '2.hdf5', g['encoded_batches'][0])
    write_encoded_batch(tmp_path / '10.hdf5', g['encoded_batches'][1])
    dl = PreprocessedDataloader(tmp_path, 3, is_raw=False)
    assert [p.name for p in dl.files] == ['2.hdf5', '10.hdf5']
    compare(next(dl), g['decoded_batch'])
    dl = PreprocessedDataloader(tmp_path, 2, is_raw=False)
    compare(next(dl), g['decoded_batches'][0])
    compare(next(dl), _third_batch(tmp_path, g))


def _third_batch(tmp_path, g):
    """Second batch of size 2 = sample 2 (file 10) + sample 0 (file 2 again)."""
    from dvs_of_training_framework_amd.preprocessed import (read_encoded_quantized_batch)
    with hdf5io.File(tmp_path / '2.hdf5', 'r') as f:
        cps = torch.from_numpy(f['channels_per_sample'][...])
        eps = torch.from_numpy(f['elements_per_sample'][...])
        first = read_encoded_quantized_batch(f, cps, eps, 0, 1)
    return enc.decode_quantized_batch(enc.join_batches([g['encoded_batches'][1], first]))


def test_compact_batches_feed_the_voxeliser_columns(tmp_path):
    """compact=True: events stay in the 9 B/event encoded columns plus
    per-sample offsets; everything else is the decoded batch."""
    from dvs_of_training_framework_amd import synthetic
    from dvs_of_training_framework_amd.preprocessed import PreprocessedDataloader
    from dvs_of_training_framework_amd.voxel import is_compact
    b = synthetic.to_torch(synthetic.make_batch(3, 4, 16, 24, 50))
    b['augmentation_params'] = {
        'idx': torch.arange(4), 'sequence_length': torch.ones(4, dtype=torch.short),
        'collapse_length': torch.ones(4, dtype=torch.short),
        'box': torch.tensor([[0, 0, 16, 24]] * 4), 'angle': torch.zeros(4),
        'is_flip': torch.zeros(4, dtype=torch.bool)}
    e = enc.encode_batch(b['events'], b['timestamps'], b['sample_idx'], b['images'],
                         b['augmentation_params'], b['size'])
    _write_parts(tmp_path, [e])
    batch = next(PreprocessedDataloader(tmp_path, 3, is_raw=True, compact=True))
    assert is_compact(batch['events']) and batch['size'] == 3
    ev = batch['events']
    assert ev['x'].dtype == torch.short and ev['polarity'].dtype == torch.bool
    assert ev['sample_event_offsets'].tolist() == [0, 50, 100, 150]
    assert torch.equal(ev['x'].long(), b['events']['x'][:150])
    assert torch.equal(batch['sample_idx'], b['sample_idx'][:6])
    assert batch['images'].dtype == torch.float32
