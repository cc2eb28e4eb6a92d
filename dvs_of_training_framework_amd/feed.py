"""Host -> device leg of the train loop, overlapped with the step before it.

The reference moves every batch with ``tensor.to(device)`` on the compute
stream at the top of ``process_minibatch`` (utils/training.py:45-56): 23 MB of
int64 event columns per batch of 8 at 256x256 (44 B/event; 4.7 MB in the
encoded 9 B/event form, utils/dataset.py:286-289) plus the frames, serialised
in front of the forward pass.  ``DeviceFeeder`` wraps a host loader and yields
DEVICE-RESIDENT batches instead:

  * ``slots`` (2) sets of device buffers of a fixed event capacity; the copy of
    batch n+1 into slot (n+1) % slots is enqueued on a COPY STREAM before
    batch n is handed to the loop, so it runs while step n computes;
  * pinned sources are copied as they are (``DataLoader(pin_memory=True)`` is
    the reference's own option, utils/dataloader.py:103-108); pageable sources
    go through the slot's own pinned staging buffers;
  * only the event columns a kernel reads travel (everything but
    ``element_index``, or the ``columns`` given);
    per-event buffers are padded to the capacity with x = y = -1 (the
    voxeliser skips those: static shapes, nothing to synchronise);
  * ordering by events, never by the host: a slot is rewritten only after the
    step that read it (``free``, recorded on the compute stream when the loop
    asks for the next batch), a batch is used only after its copy (``ready``).

The yielded dict has the reference's keys (so ``process_minibatch`` / ``train``
take it unchanged: ``.to(device)`` is the identity on device tensors) plus
``slot`` (whose buffers these are: ``capture.CapturedLoop`` binds one captured
step per slot to them -- no staging copy between feeder and step) and
``num_events`` (events before padding: the oversize-batch rule of
utils/training.py:141-150 counts those).
"""
import torch

from . import voxel

WIRE_COLUMNS = ('x', 'y', 'timestamp', 'polarity', 'element_index', 'sample_index')
_COPY_STREAMS = {}      # device index -> THE copy stream of that device


def copy_stream(device):
    """One copy stream per device for every feeder of the process: ROCclr binds
    each new stream to one of GPU_MAX_HW_QUEUES hardware queues, and a stream
    per feeder would sooner or later share a queue with the compute streams
    (the copies would then wait behind kernels; see parallel.claim_streams)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _COPY_STREAMS[key]


def _pow2_at_least(n, floor=4096):
    return max(floor, 1 << max(int(n) - 1, 1).bit_length())


class _Slot:
    def __init__(self, index):
        self.index = index
        self.device = {}        # name -> device tensor ('events' nested)
        self.pinned = {}        # name -> pinned staging tensor
        self.capacity = 0
        self.valid = 0          # events of the batch the buffers hold
        self.free = None        # compute stream is done reading the slot
        self.ready = torch.cuda.Event()
        self.generation = 0     # bumped when the buffers are re-allocated


class DeviceFeeder:
    def __init__(self, loader, device, slots=2, event_capacity=None,
                 columns=None, copy_stream=None):
        """loader: iterable of host batches in the reference's wire format (or
        with compact event columns, voxel.is_compact).  columns: event columns
        to move (None: every column of the batch except ``element_index``,
        which no kernel of the model reads)."""
        self.loader, self.device = loader, torch.device(device)
        # batch n+1 is staged BEFORE batch n is handed out: with one slot it would overwrite
        # the batch the loop is about to train on
        assert slots >= 2, 'DeviceFeeder needs at least two slots'
        self.slots = [_Slot(i) for i in range(slots)]
        self.event_capacity = event_capacity
        self.columns = columns
        self.stream = copy_stream or globals()['copy_stream'](self.device)
        self.bytes_moved = 0
        self.batches = 0

    def __len__(self):
        return len(self.loader)

    # ------------------------------------------------------------------
    def _columns(self, events):
        if self.columns is not None:
            return [k for k in self.columns if k in events]
        return [k for k in events if k not in ('size', 'element_index')]

    def _alloc(self, slot, batch, cols, n):
        ev = batch['events']
        cap = max(self.event_capacity or 0, _pow2_at_least(n))
        dev = self.device
        per_event = [k for k in cols if k != 'sample_event_offsets']
        slot.device = {
            'events': {k: torch.empty(cap if k in per_event else ev[k].numel(),
                                      dtype=ev[k].dtype, device=dev) for k in cols},
            'timestamps': torch.empty_like(batch['timestamps'], device=dev),
            'sample_idx': torch.empty_like(batch['sample_idx'], device=dev),
            'images': torch.empty_like(batch['images'], device=dev)}
        for k in ('x', 'y'):
            slot.device['events'][k].fill_(-1)
        if 'sample_index' in slot.device['events']:
            slot.device['events']['sample_index'].zero_()
        slot.pinned = {}
        slot.capacity, slot.valid = cap, 0
        slot.shape = self._shape(batch)
        slot.generation += 1
        torch.cuda.current_stream(dev).synchronize()    # fills done before the copy stream writes

    @staticmethod
    def _shape(batch):
        return (tuple(batch['images'].shape), tuple(batch['timestamps'].shape),
                voxel.is_compact(batch['events']))

    def _pinned_source(self, slot, name, t, per_event=False):
        """``t`` itself when it is pinned, else the slot's staging copy."""
        if t.is_pinned():
            return t
        buf = slot.pinned.get(name)
        if buf is None or buf.numel() < t.numel() or buf.dtype != t.dtype:
            cap = max(t.numel(), slot.capacity if per_event else 0)
            buf = slot.pinned[name] = torch.empty(cap, dtype=t.dtype).pin_memory()
        view = buf[:t.numel()].view(t.shape)
        view.copy_(t)
        return view

    def _stage(self, batch, slot):
        """Enqueue the copy of ``batch`` into ``slot`` on the copy stream."""
        ev = batch['events']
        cols = self._columns(ev)
        n = ev['x'].numel()
        if not slot.device or n > slot.capacity or slot.shape != self._shape(batch) \
                or set(cols) != set(slot.device['events']):
            if slot.free is not None:
                slot.free.synchronize()
            self._alloc(slot, batch, cols, n)
        st = self.stream
        if slot.free is not None:
            st.wait_event(slot.free)            # the step that read the slot has finished
        if any(not t.is_pinned() for t in [ev[k] for k in cols] + [batch['images']]):
            # the staging buffers are rewritten by the host: the previous copy out of them
            # must have completed
            slot.ready.synchronize()
        moved = 0
        with torch.cuda.stream(st):
            dst = slot.device['events']
            for k in cols:
                src = self._pinned_source(slot, 'events.' + k, ev[k], k != 'sample_event_offsets')
                if k == 'sample_event_offsets':
                    dst[k].copy_(src, non_blocking=True)
                else:
                    dst[k][:n].copy_(src, non_blocking=True)
                moved += src.numel() * src.element_size()
            if n < slot.valid:                  # the previous batch was longer: re-pad its tail
                dst['x'][n:slot.valid].fill_(-1)
                dst['y'][n:slot.valid].fill_(-1)
            for k in ('timestamps', 'sample_idx', 'images'):
                src = self._pinned_source(slot, k, batch[k])
                slot.device[k].copy_(src, non_blocking=True)
                moved += src.numel() * src.element_size()
            slot.ready.record(st)
        slot.valid = n
        self.bytes_moved += moved
        self.batches += 1
        out = {'events': dict(slot.device['events']), 'timestamps': slot.device['timestamps'],
               'sample_idx': slot.device['sample_idx'], 'images': slot.device['images'],
               'size': batch['size'], 'num_events': n, 'slot': (slot.index, slot.generation),
               'augmentation_params': batch.get('augmentation_params')}
        return out, slot

    def __iter__(self):
        it = iter(self.loader)
        k = 0
        try:
            nxt = self._stage(next(it), self.slots[0])
        except StopIteration:
            return
        while nxt is not None:
            cur, slot = nxt
            k += 1
            try:        # batch n+1 is on its way before batch n is handed out
                nxt = self._stage(next(it), self.slots[k % len(self.slots)])
            except StopIteration:
                nxt = None
            main = torch.cuda.current_stream(self.device)
            main.wait_event(slot.ready)
            yield cur
            # the loop is back for the next batch: everything that reads `slot` is enqueued
            if slot.free is None:
                slot.free = torch.cuda.Event()
            slot.free.record(torch.cuda.current_stream(self.device))
