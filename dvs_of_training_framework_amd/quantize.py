"""Offline voxelisation of a dataset: the counterpart of the reference's
``scripts/quantize_preprocessed.py`` (main loop :60-113) on top of the HIP
voxeliser.

For every batch of the loader the event columns are replaced by
``data = model.quantize(events, timestamps, sample_idx, imsize)`` (:87-91),
the batch is put in ``encode_quantized_batch`` form (:93), groups of
``ceil(samples_per_file / mbs)`` batches are joined (``join_batches``, :100)
and handed to ``sink(file_index, joined)`` -- the reference writes
``{j}.hdf5`` with ``write_encoded_batch`` (:104); HDF5 I/O is outside this
package (h5py is not a dependency), so the writer is a parameter and
``torch_file_sink`` stores ``{j}.pt`` files instead.  A dataset produced this
way is consumed by the ``raw=False`` training path
(``training.process_minibatch(..., is_raw=False)``, reference
utils/training.py:54-55) through ``decode_quantized_batch``.
"""
import copy
from pathlib import Path

import torch

from .encoding import (decode_quantized_batch, encode_quantized_batch,
                       join_batches)


def _to_device(batch, device):
    """The H2D block of the reference loop (:78-86)."""
    for k in set(batch['events'].keys()) - {'size'}:
        batch['events'][k] = batch['events'][k].to(device)
    for k in set(batch.keys()) - {'events', 'size', 'augmentation_params'}:
        batch[k] = batch[k].to(device)
    for k in set(batch['augmentation_params'].keys()):
        batch['augmentation_params'][k] = \
            batch['augmentation_params'][k].to(device)
    return batch


def quantize_batch(model, batch, device):
    """One wire-format batch -> encoded quantized batch (host tensors)."""
    imsize = batch['images'].size()[-2:]
    batch = _to_device(batch, device)
    quantized = copy.copy(batch)
    quantized['augmentation_params'] = dict(batch['augmentation_params'])
    del quantized['events']
    with torch.no_grad():
        kwargs = {}
        if 'size' in batch:
            kwargs['batch_size'] = int(batch['size'])
        quantized['data'] = model.quantize(batch['events'],
                                           batch['timestamps'],
                                           batch['sample_idx'], imsize,
                                           **kwargs)
    return encode_quantized_batch(quantized)


def torch_file_sink(output):
    """sink writing ``{output}/{j}.pt`` (``torch.save`` of the joined dict)."""
    output = Path(output)
    output.mkdir(exist_ok=True, parents=True)

    def sink(j, joined):
        torch.save(joined, output / f'{j}.pt')
    return sink


def quantize_dataset(loader, model, device, sink, mbs, samples_per_file,
                     size=None, num_written=0, written_indices=()):
    """Reference scripts/quantize_preprocessed.py:60-113.

    loader            iterable of wire-format batches of ``mbs`` samples
    sink(j, joined)   receives every joined group of encoded batches
    size              stop after this many samples (None: whole loader)
    num_written / written_indices
                      resume state: samples already on disk and the file
                      indices they occupy (never reused)
    -> number of samples written in total
    """
    num_batches_per_write = (samples_per_file - 1) // mbs + 1
    written_indices = set(written_indices)
    encoded_batches, j = [], 0
    for i, batch in enumerate(loader):
        if size is not None and num_written >= size:
            break
        encoded_batches.append(quantize_batch(model, batch, device))
        num_written += len(encoded_batches[-1]['elements_per_sample'])
        is_last = size is not None and num_written >= size
        if (i + 1) % num_batches_per_write == 0 or is_last:
            joined = join_batches(encoded_batches)
            while j in written_indices:
                j += 1
            sink(j, joined)
            j += 1
            encoded_batches = []
        if is_last:
            break
    if encoded_batches:                 # loader ended before a write boundary
        while j in written_indices:
            j += 1
        sink(j, join_batches(encoded_batches))
    return num_written


def iterate_quantized(joined, mbs):
    """Joined encoded quantized batch -> decoded batches of ``mbs`` samples
    ({'data','timestamps','sample_idx','images','augmentation_params',
    'size'}), the form ``process_minibatch(is_raw=False)`` takes."""
    from .encoding import select_quantized_ranges
    n = len(joined['elements_per_sample'])
    for b in range(0, n, mbs):
        e = min(n, b + mbs)
        r = select_quantized_ranges(joined['channels_per_sample'],
                                    joined['elements_per_sample'], b, e)

        def cut(v, rr):
            if v is None:
                return None
            if isinstance(rr, dict) and 'begin' not in rr:
                return {k: cut(v[k], rr[k]) for k in rr if k in v}
            return v[rr['begin']:rr['end']]
        part = {k: cut(joined[k], r[k]) for k in r}
        yield decode_quantized_batch(part)
