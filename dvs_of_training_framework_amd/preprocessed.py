"""Preprocessed (encoded / quantized) datasets on disk and the loader over
them -- the surface of the reference's ``utils/dataset.py``:

  write_encoded_batch            :376-394   nested dict of tensors -> HDF5
  read_data                      :505-518   {'begin','end'} range tree -> tensors
  read_encoded_batch             :397-426
  read_encoded_quantized_batch   :521-548
  PreprocessedDataloader         :799-954   files ``<int>.hdf5`` in numeric
        order, cycled endlessly; a batch may span files (and wrap around to
        the first file); ``set_index`` fast-forwards to a sample (resume,
        train_flownet.py:188-190); ``<stem>.info`` caches a file's size.

pinned by tests/dataset/test_encoding.py:270-313 and
tests/dataset/test_quantization.py:145-166 (restated on the committed
literals in tests/test_preprocessed.py).  The file backend is ``hdf5io``
(libhdf5 through ctypes; h5py is not installed here), the files are h5py's.

MI355X addition: ``compact=True`` hands the raw events over as the 9 B/event
encoded columns + per-sample offsets (``encoding.compact_events``), which
``Model.forward`` voxelises on the device without ever building the 44 B/event
int64 wire columns -- the "streaming voxelisation" of BASELINE configs[4].

The reference's caching file iterators (utils/file_iterators.py: a prefetch
thread copying files to fast storage) are host-side I/O outside this build's
scope; ``cache_dir`` is accepted and ignored with a warning.
"""
import logging
from pathlib import Path

import torch
import yaml

from . import hdf5io
from .encoding import (compact_events, decode_batch, decode_batch_info,
                       decode_quantized_batch, join_batches,
                       select_encoded_ranges, select_quantized_ranges)

log = logging.getLogger(__name__)


# ------------------------------------------------------------------ file io
def write_encoded_batch(path, batch):
    """Nested dict of tensors -> groups / datasets of the same names."""
    def put(group, name, value):
        if isinstance(value, torch.Tensor):
            group.create_dataset(name, data=value)
            return
        assert isinstance(value, dict), name
        sub = group.create_group(name)
        for k, v in value.items():
            put(sub, k, v)
    with hdf5io.File(path, 'w') as f:
        for k, v in batch.items():
            put(f, k, v)


def _is_range(node):
    assert isinstance(node, dict), node
    return isinstance(node.get('begin'), int) and isinstance(node.get('end'), int)


def read_data(descriptor, ranges):
    """Mirror ``ranges`` (a tree whose leaves are {'begin', 'end'}) with the
    rows ``[begin, end)`` of the datasets of the same names."""
    assert isinstance(ranges, dict)
    out = {}
    for name, node in ranges.items():
        if _is_range(node):
            out[name] = torch.from_numpy(
                descriptor[name][node['begin']:node['end']])
        else:
            out[name] = read_data(descriptor[name], node)
    return out


def _column(descriptor, *path):
    node = descriptor
    for p in path:
        node = node[p]
    return torch.from_numpy(node[...])


def read_encoded_batch(descriptor, events_per_element, elements_per_sample,
                       sample_begin, sample_end):
    return read_data(descriptor, select_encoded_ranges(
        events_per_element, elements_per_sample, sample_begin, sample_end))


def read_encoded_quantized_batch(descriptor, channels_per_sample,
                                 elements_per_sample, sample_begin,
                                 sample_end):
    return read_data(descriptor, select_quantized_ranges(
        channels_per_sample, elements_per_sample, sample_begin, sample_end))


# ---------------------------------------------------------------- file cycle
class _Handle:
    """What the reference's file iterators hand out: ``.name`` and
    ``.release()`` (utils/file_iterators.py:18-60)."""

    def __init__(self, path):
        self.name = Path(path)

    def release(self):
        pass


class FileCycle:
    """Endless in-order iteration over the files (FileIterator,
    utils/file_iterators.py:96-121)."""

    def __init__(self, files):
        self.files = [Path(f) for f in files]
        self.index = 0

    def next(self, blocking=True):
        f = self.files[self.index]
        self.index = (self.index + 1) % len(self.files)
        return _Handle(f)

    def reset(self):
        self.index = 0


# -------------------------------------------------------------------- loader
class PreprocessedDataloader:
    """Sequential batches out of a directory of preprocessed files.

    Attributes (as upstream): file_index / sample_index (position of the next
    sample: current file, index inside it), batch_size, files, length.
    """

    def __init__(self, path, batch_size, is_raw, cache_dir=None, cache_size=0,
                 process_only_once=True, compact=False):
        path = Path(path)
        self.batch_size = batch_size
        self.is_raw = is_raw
        self.compact = bool(compact) and is_raw
        self.files = sorted(path.glob('*.hdf5'), key=lambda p: int(p.stem))
        assert len(self.files) > 0, \
            f'No preprocessed dataset at {path} (no .hdf5 files)'
        if cache_dir is not None:
            log.warning('cache_dir is ignored: files are read in place')
        self.iterator = FileCycle(self.files)
        self._sizes = {}
        self.length = sum(self._file2size(f, save_info=True)
                          for f in self.files)
        self.sample_index = 0
        self.current_file = self.iterator.next()

    # ---- sizes
    @staticmethod
    def _hdf5file2size(filename):
        with hdf5io.File(filename, 'r') as f:
            return len(f['elements_per_sample'])

    def _file2size(self, filename, save_info=False):
        """Samples in a file; ``<stem>.info`` (yaml ``size:``) short-cuts the
        open and is written on first contact."""
        filename = Path(filename)
        if filename in self._sizes:
            return self._sizes[filename]
        info = filename.with_suffix('.info')
        if info.is_file():
            size = yaml.safe_load(info.read_text())['size']
        else:
            size = self._hdf5file2size(filename)
            if save_info:
                info.write_text(yaml.dump({'size': size}))
        self._sizes[filename] = size
        return size

    # ---- position
    def set_index(self, idx):
        """Continue from sample ``idx`` (modulo the dataset length)."""
        self.sample_index = idx % self.length
        self.current_file.release()
        self.iterator.reset()
        self.current_file = self.iterator.next()
        while self.sample_index >= self._file2size(self.current_file.name):
            self.sample_index -= self._file2size(self.current_file.name)
            self.current_file.release()
            self.current_file = self.iterator.next()

    def __len__(self):
        return self.length

    def __iter__(self):
        return self

    # ---- reading
    @staticmethod
    def _read_raw_batch(descriptor, begin, end):
        return read_encoded_batch(
            descriptor, _column(descriptor, 'events', 'events_per_element'),
            _column(descriptor, 'elements_per_sample'), begin, end)

    @staticmethod
    def _read_quantized_batch(descriptor, begin, end):
        return read_encoded_quantized_batch(
            descriptor, _column(descriptor, 'channels_per_sample'),
            _column(descriptor, 'elements_per_sample'), begin, end)

    def _decode(self, encoded):
        if not self.is_raw:
            return decode_quantized_batch(encoded)
        if not self.compact:
            return decode_batch(encoded)
        batch = decode_batch_info(encoded)
        batch['events'] = compact_events(encoded)
        return batch

    def __next__(self):
        read = self._read_raw_batch if self.is_raw \
            else self._read_quantized_batch
        wanted, parts = self.batch_size, []
        while wanted > 0:
            here = self._file2size(self.current_file.name) - self.sample_index
            take = min(here, wanted)
            if take > 0:
                with hdf5io.File(self.current_file.name, 'r') as f:
                    parts.append(read(f, self.sample_index,
                                      self.sample_index + take))
            self.sample_index += take
            wanted -= take
            if wanted > 0:          # next file (after the last one: the first)
                self.current_file.release()
                self.current_file = self.iterator.next()
                self.sample_index = 0
        return self._decode(join_batches(parts))
