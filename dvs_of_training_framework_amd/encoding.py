"""Encoded (raw-event) and quantized on-disk batch formats.

Same functions, argument meaning, dtypes and dictionary layouts as the
reference's ``utils/dataset.py`` (select_*_ranges :28-156, join_batches
:159-198, encode/decode_batch(_info) :201-373, encode/decode_quantized_batch
:429-502); pinned bit-exact by the literal goldens of
``tests/dataset/test_encoding.py`` and ``tests/dataset/test_quantization.py``
(extracted as data into tests/golden/encoding.pt).  The HDF5 container of
these dictionaries is ``hdf5io`` (libhdf5 through ctypes; h5py is absent),
the loader over it ``preprocessed.PreprocessedDataloader``.

The encoded event columns (int16 x, int16 y, float32 timestamp, bool
polarity = 9 bytes/event) can be voxelised on the GPU without expanding them
to the 44-byte int64 wire format: ``voxelize_encoded``.
"""
import torch


def cumsum_with_prefix(tensor, dtype=None):
    """[1, 2, 3] -> [0, 1, 3, 6] (reference utils/common.py:26-50)."""
    dtype = tensor.dtype if dtype is None else dtype
    out = torch.zeros(tensor.numel() + 1, dtype=dtype, device=tensor.device)
    out[1:] = torch.cumsum(tensor, dim=0)
    return out


_AUG_KEYS = ('idx', 'sequence_length', 'collapse_length', 'box', 'angle',
             'is_flip')


def _span(b, e):
    return {'begin': b, 'end': e}


def select_batch_info_ranges(elements_per_sample, sample_begin, sample_end):
    assert isinstance(sample_begin, int) and isinstance(sample_end, int)
    assert sample_end > sample_begin
    # sample i owns elements_per_sample[i] + 1 timestamps / frames
    ts_shift = cumsum_with_prefix(elements_per_sample.to(torch.long) + 1)
    tb, te = int(ts_shift[sample_begin]), int(ts_shift[sample_end])
    return {'timestamps': _span(tb, te),
            'elements_per_sample': _span(sample_begin, sample_end),
            'images': _span(tb, te),
            'augmentation_params': {k: _span(sample_begin, sample_end)
                                    for k in _AUG_KEYS}}


def select_encoded_ranges(events_per_element, elements_per_sample,
                          sample_begin, sample_end):
    assert isinstance(sample_begin, int) and isinstance(sample_end, int)
    assert sample_end > sample_begin
    ev_shift = cumsum_with_prefix(events_per_element.to(torch.long))
    el_shift = cumsum_with_prefix(elements_per_sample.to(torch.long))
    eb, ee = int(el_shift[sample_begin]), int(el_shift[sample_end])
    vb, ve = int(ev_shift[eb]), int(ev_shift[ee])
    result = select_batch_info_ranges(elements_per_sample, sample_begin,
                                      sample_end)
    result['events'] = {k: _span(vb, ve)
                        for k in ('x', 'y', 'timestamp', 'polarity')}
    result['events']['events_per_element'] = _span(eb, ee)
    return result


def select_quantized_ranges(channels_per_sample, elements_per_sample,
                            sample_begin, sample_end):
    assert isinstance(sample_begin, int) and isinstance(sample_end, int)
    assert sample_end > sample_begin
    ch_shift = cumsum_with_prefix(channels_per_sample.to(torch.long))
    result = select_batch_info_ranges(elements_per_sample, sample_begin,
                                      sample_end)
    result['data'] = _span(int(ch_shift[sample_begin]),
                           int(ch_shift[sample_end]))
    result['channels_per_sample'] = _span(sample_begin, sample_end)
    return result


def join_batches(batches):
    """Concatenate encoded batches field by field."""
    if len(batches) == 0:
        empty = lambda dt: torch.tensor([], dtype=dt)   # noqa: E731
        return {'events': {'x': empty(torch.short), 'y': empty(torch.short),
                           'timestamp': empty(torch.float32),
                           'polarity': empty(torch.bool),
                           'events_per_element': empty(torch.short)},
                'timestamps': empty(torch.float32),
                'elements_per_sample': empty(torch.short),
                'images': empty(torch.uint8),
                'augmentation_params': {}}
    if len(batches) == 1:
        return batches[0]
    first, out = batches[0], {}
    for key, val in first.items():
        if isinstance(val, dict):
            out[key] = {sk: torch.cat([b[key][sk] for b in batches])
                        for sk in val}
        elif val is None:
            assert key == 'augmentation_params'
            assert all(b[key] is None for b in batches)
            out[key] = None
        else:
            assert isinstance(val, torch.Tensor)
            out[key] = torch.cat([b[key] for b in batches])
    return out


def encode_batch_info(timestamps, sample_idx, images, augmentation_params,
                      size):
    # a sample with k timestamps has k - 1 elements
    counts = torch.bincount(sample_idx.to(torch.long), minlength=size)
    return {'timestamps': timestamps,
            'elements_per_sample': (counts - 1).to(torch.uint8),
            'images': images.to(torch.uint8),
            'augmentation_params': augmentation_params}


def encode_batch(events, timestamps, sample_idx, images, augmentation_params,
                 size):
    result = encode_batch_info(timestamps, sample_idx, images,
                               augmentation_params, size)
    el_shift = cumsum_with_prefix(result['elements_per_sample'].to(torch.long))
    # global element id of every event; the table ends at the LAST event's
    # element (trailing empty elements are not stored, as in the reference)
    gid = events['element_index'].to(torch.long) + \
        el_shift[events['sample_index'].to(torch.long)]
    total = int(gid[-1]) + 1
    result['events'] = {
        'x': events['x'].to(torch.short),
        'y': events['y'].to(torch.short),
        'timestamp': events['timestamp'],
        'polarity': ((events['polarity'] + 1) / 2).to(torch.bool),
        'events_per_element': torch.bincount(gid, minlength=total)}
    return result


def decode_batch_info(encoded):
    eps = encoded['elements_per_sample'].to(torch.long)
    return {'timestamps': encoded['timestamps'].to(torch.float32),
            'sample_idx': torch.repeat_interleave(
                torch.arange(eps.numel(), dtype=torch.long), eps + 1),
            'images': encoded['images'].to(torch.float32),
            'augmentation_params': encoded['augmentation_params'],
            'size': eps.numel()}


def decode_batch(encoded):
    result = decode_batch_info(encoded)
    ev = encoded['events']
    eps = encoded['elements_per_sample'].to(torch.long)
    epe = ev['events_per_element'].to(torch.long)
    # element ids inside their sample, one per stored element
    el_shift = cumsum_with_prefix(eps)
    el_sample = torch.repeat_interleave(torch.arange(eps.numel()), eps)
    el_local = torch.arange(int(eps.sum())) - el_shift[el_sample]
    n_stored = epe.numel()
    result['events'] = {
        'x': ev['x'].to(torch.long),
        'y': ev['y'].to(torch.long),
        'timestamp': ev['timestamp'],
        'polarity': ev['polarity'].to(torch.long) * 2 - 1,
        'element_index': torch.repeat_interleave(el_local[:n_stored], epe),
        'sample_index': torch.repeat_interleave(el_sample[:n_stored], epe)}
    return result


def encode_quantized_batch(batch):
    def cpu(v):
        if isinstance(v, dict):
            return {k: cpu(x) for k, x in v.items()}
        return v.cpu() if isinstance(v, torch.Tensor) else v
    batch = cpu(batch)
    B, C, H, W = batch['data'].size()
    result = {'data': batch['data'].reshape(B * C, H, W),
              'channels_per_sample': torch.full((B,), C, dtype=torch.uint8)}
    result.update(encode_batch_info(batch['timestamps'], batch['sample_idx'],
                                    batch['images'],
                                    batch['augmentation_params'],
                                    batch['size']))
    return result


def decode_quantized_batch(batch):
    result = decode_batch_info(batch)
    cps = batch['channels_per_sample']
    assert cps.numel() > 0
    assert bool((cps == cps[0]).all())
    C = int(cps[0])
    _, H, W = batch['data'].size()
    result['data'] = batch['data'].view(result['size'], C, H, W)
    return result


def sample_event_offsets(encoded):
    """int64[B+1]: first event of every sample in the encoded event columns."""
    ev_shift = cumsum_with_prefix(
        encoded['events']['events_per_element'].to(torch.long))
    el_shift = cumsum_with_prefix(
        encoded['elements_per_sample'].to(torch.long))
    # elements past the stored table hold no events
    el_shift = el_shift.clamp(max=ev_shift.numel() - 1)
    return ev_shift[el_shift]


def compact_events(encoded):
    """Encoded batch -> the compact event dict ``Model.forward`` accepts in
    place of the int64 wire columns (voxel.is_compact): the four encoded
    columns as stored plus ``sample_event_offsets``."""
    ev = encoded['events']
    return {'x': ev['x'], 'y': ev['y'], 'timestamp': ev['timestamp'],
            'polarity': ev['polarity'],
            'sample_event_offsets': sample_event_offsets(encoded)}


def voxelize_encoded(encoded, t0, t1, C, H, W, device='cuda', debug=False):
    """Encoded batch (host or device tensors) -> float32 grid [B,C,H,W] with
    the arithmetic of docs/VOXEL_SPEC.md, reading the 9 B/event columns."""
    from .voxel import voxelize_compact
    dev = torch.device(device)
    ev = compact_events(encoded)
    B = ev['sample_event_offsets'].numel() - 1
    return voxelize_compact(ev, t0.to(dev), t1.to(dev), B, C, H, W, debug)
