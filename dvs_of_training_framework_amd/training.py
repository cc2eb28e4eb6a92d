"""Train / validate loops behind the reference's surface (utils/training.py).

Public names, signatures, return structures, hook protocol and TensorBoard
tags are the reference's: ``combined_loss`` (:12-24), ``make_hook_periodic``
(:27-30), ``process_minibatch`` (:37-86), ``train`` (:89-235), ``validate``
(:244-271).  The organisation underneath is this build's:

  * ``TermReadback``   the 3 x n_scales loss terms of one micro-batch as host
                       floats, fetched by ONE packed device->host copy at first
                       use (the reference makes 12 ``.item()`` syncs, :77);
  * ``ScaleSums``      per-scale running sums between optimizer steps and the
                       scalar tags they are written under;
  * ``_StepClock``     micro-batch / optimizer-step / sample counters and the
                       oversize-batch bookkeeping (:141-150);
  * ``train``          drives them; ``reducer`` (parallel.GradReducer) adds
                       the data-parallel gradient exchange: buckets are
                       all-reduced on a side stream during backward and joined
                       before ``optimizer.step()``; micro-batches that only
                       accumulate do not exchange.
"""
import torch

from .common import mean
from .loss import unit_backward
from .timer import EventTimer, FakeTimer

TERM_NAMES = ('smoothness', 'photometric', 'out_reg')       # order of :171
STAGES = ('batch_construction', 'batch2gpu', 'forward', 'loss', 'backprop',
          'optimizer_step', 'logging')


# --------------------------------------------------------------------- loss
class _Terms(tuple):
    """((smooth_k), (photo_k), (border_k)) of 0-dim tensors that are views of
    one [3,K] tensor (``packed``): host readers fetch it in one copy."""
    def __new__(cls, packed):
        self = super().__new__(cls, (tuple(r.unbind(0))
                                     for r in packed.unbind(0)))
        self.packed = packed
        return self


def combined_loss(evaluator, flows, flow_ts, flow_sample_idx, images,
                  timestamps, sample_idx, features, weights=[0.5, 1, 1],
                  frame_indices=None):
    """-> (sum_t weights[t] * mean_k term[t][k], terms).  With an evaluator
    that has ``fused`` and flows that need gradients the value and the flow
    gradients come out of one sweep (loss.Losses.fused)."""
    extra = {} if frame_indices is None else {'frame_indices': frame_indices}
    call = (flows, flow_ts, flow_sample_idx, images, timestamps, sample_idx)
    trainable = torch.is_grad_enabled() and \
        any(f.requires_grad for f in flows)
    if trainable and hasattr(evaluator, 'fused'):
        loss, packed = evaluator.fused(*call, weights=weights, **extra)
        return loss, _Terms(packed)
    terms = evaluator(*call, **extra)
    loss = 0
    for weight, per_scale in zip(weights, terms):
        loss = loss + weight * mean(per_scale)
    return loss, terms


class TermReadback:
    """Host view of the loss terms of one micro-batch.

    Iterating yields one iterator of floats per term (smoothness,
    photometric, out-of-border) -- the structure ``process_minibatch``
    returns in the reference -- and ``host()`` the whole table.  Nothing is
    copied until a value is asked for; then everything is, once."""

    def __init__(self, terms):
        self._terms = terms
        self.n_terms = len(terms)
        self.n_scales = len(terms[0]) if self.n_terms else 0
        self._table = None

    def host(self):
        if self._table is None:
            packed = getattr(self._terms, 'packed', None)
            if packed is None:
                packed = torch.stack([torch.stack(list(t))
                                      for t in self._terms])
            self._table = packed.detach().cpu().tolist()
            self._terms = None
        return self._table

    def row(self, t):
        """Floats of term ``t``, one per scale (a generator: still lazy)."""
        return (self.host()[t][k] for k in range(self.n_scales))

    def __len__(self):
        return self.n_terms

    def __iter__(self):
        # row(t) binds t per call: unpacking all rows first and reading them
        # afterwards gives three different rows
        return iter([self.row(t) for t in range(self.n_terms)])


def _lazy_items(terms):
    return TermReadback(terms)


def make_hook_periodic(hook, checkpointing_interval):
    def periodic(step, *args):
        if step % checkpointing_interval == 0:
            return hook(step, *args)
        return None
    return periodic


def predictions2tag(predictions):
    return (f'{p.shape[-2]}x{p.shape[-1]}' for p in predictions)


# ---------------------------------------------------------------- one batch
def _to_device(batch, device, is_raw):
    def put(t):
        return t.to(device, non_blocking=True)
    frames = tuple(put(batch[k]) for k in ('timestamps', 'sample_idx',
                                           'images'))
    if not is_raw:
        return frames + (put(batch['data']),)
    events = batch['events']
    for name in [k for k in events if k != 'size']:
        events[name] = put(events[name])
    return frames + (events,)


def process_minibatch(model, batch, timers, device, is_raw, evaluator,
                      weights, return_prediction=False):
    """batch (SURVEY 8b wire format) -> (loss, terms, tags[, outputs])."""
    with _timed(timers, 'batch2gpu'):
        timestamps, sample_idx, images, payload = _to_device(batch, device,
                                                             is_raw)
    with _timed(timers, 'forward'):
        hints = {}
        if hasattr(model, 'last_frame_indices') and 'size' in batch:
            hints['batch_size'] = int(batch['size'])    # B without a device sync
        flows, flow_ts, flow_sample_idx, features = model(
            payload, timestamps, sample_idx, images.size()[-2:], raw=is_raw,
            intermediate=True, **hints)
        tags = predictions2tag(flows)
    with _timed(timers, 'loss'):
        loss, terms = combined_loss(
            evaluator, flows, flow_ts, flow_sample_idx, images, timestamps,
            sample_idx, features, weights=weights,
            frame_indices=getattr(model, 'last_frame_indices', None))
        terms = TermReadback(terms)
    if not return_prediction:
        return loss, terms, tags
    return loss, terms, tags, {'prediction': flows, 'flow_ts': flow_ts,
                               'flow_sample_idx': flow_sample_idx,
                               'features': features}


class _timed:
    """``with _timed(timers, name):`` = timers(name).start() / .stop()."""
    def __init__(self, timers, name):
        self.t = timers(name)

    def __enter__(self):
        self.t.start()

    def __exit__(self, *exc):
        self.t.stop()
        return False


# ------------------------------------------------------------- bookkeeping
class ScaleSums:
    """Per-scale sums of the three terms and of the loss since the last
    optimizer step (train) or over a whole pass (validate)."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.per_term = None        # [term][scale]
        self.loss = 0.0
        self.tags = None

    def add(self, loss, readback, tags):
        rows = [list(r) for r in readback]
        if self.per_term is None:
            self.per_term = rows
        else:
            self.per_term = [[a + b for a, b in zip(acc, new)]
                             for acc, new in zip(self.per_term, rows)]
        self.loss += loss.item()
        self.tags = list(tags)

    def write(self, logger, x, divisor, families):
        """families: per term the tag prefix, in TERM_NAMES order."""
        for k, tag in enumerate(self.tags):
            for t, family in families:
                logger.add_scalar(f'{family}/{tag}',
                                  self.per_term[t][k] / divisor, x)


_TRAIN_FAMILIES = ((1, 'Train/photometric loss'), (0, 'Train/smoothness loss'),
                   (2, 'Train/out regularization'))
_VALID_FAMILIES = ((0, 'Validation/smoothness loss'),
                   (1, 'Validation/photometric loss'),
                   (2, 'Validation/out regularization loss'))


class _StepClock:
    """Counts micro-batches (``micro``), derives optimizer steps and keeps
    ``samples_passed``; knows which micro-batch closes an optimizer step."""

    def __init__(self, init_step, accumulation_steps, num_steps,
                 init_samples_passed):
        self.accum = accumulation_steps
        self.first = self.micro = init_step * accumulation_steps
        self.last = num_steps * accumulation_steps
        self.samples = init_samples_passed
        self.skipped = 0

    def finished(self):
        return self.micro == self.last

    def admit(self, batch):
        self.micro += 1
        self.samples += batch['size']
        return self.micro % self.accum == 0

    @property
    def step(self):
        return self.micro // self.accum

    def skip(self, num_events, batch):
        self.skipped += 1
        done = self.micro - self.first
        print(f'batch of {num_events} events exceeds max_events_per_batch: '
              f'skipped (augmentation {batch.get("augmentation_params")}); '
              f'{done / (done + self.skipped):.2f} of the batches seen so far '
              'were processed')


# -------------------------------------------------------------------- loops
def train(model, device, loader, optimizer, num_steps: int, scheduler, logger,
          evaluator, weights=[0.5, 1, 1], is_raw=True, accumulation_steps=1,
          timers=None, hooks={}, init_step=0, init_samples_passed=0,
          max_events_per_batch: int = 350000, reducer=None,
          log_every: int = 1, capture=False):
    """Semantics of utils/training.py:89-235: ``accumulation_steps``
    micro-batches per optimizer step (each loss scaled by 1/accumulation_steps),
    batches above ``max_events_per_batch`` events are skipped and not counted,
    optimizer + scheduler step on the boundary, per-scale scalars against
    ``samples_passed``, then ``hooks[name](step, samples_passed)``.

    reducer:   parallel.GradReducer (data parallelism)
    log_every: write scalars every n-th optimizer step (1 = the reference)
    capture:   replay the loop body from ONE C call per micro-batch
               (capture.CapturedLoop / the step executor) once a batch
               signature has been seen: raw events (wire or compact columns),
               with or without accumulation, with or without a reducer (the
               gradient exchange is then issued by the executor).  A batch of
               another signature runs the same body eagerly; more events than
               the captured buffers hold re-records at a larger capacity; if
               recording fails training continues eagerly
    """
    if timers is None:
        on_gpu = torch.device(device).type == 'cuda'
        timers = EventTimer() if on_gpu else FakeTimer()
    clock = _StepClock(init_step, accumulation_steps, num_steps,
                       init_samples_passed)
    sums = ScaleSums()
    captured = None
    if capture and is_raw and hasattr(optimizer, 'begin_capture'):
        from .capture import CapturedLoop
        captured = CapturedLoop(model, evaluator, optimizer, weights, device,
                                accumulation_steps, reducer)
    model.train()
    optimizer.zero_grad(set_to_none=True)

    timers('batch_construction').start()
    for batch in loader:
        if clock.finished():
            break
        if is_raw:
            # (a feed.DeviceFeeder batch is padded to its slot's capacity)
            num_events = batch.get('num_events', batch['events']['x'].numel())
            if num_events > max_events_per_batch:
                clock.skip(num_events, batch)
                continue
        timers('batch_construction').stop()
        closes_step = clock.admit(batch)
        if reducer is not None:
            reducer.enabled = closes_step
        if hasattr(optimizer, 'fused_active'):      # optim.fuse_into_backward
            optimizer.fused_active = closes_step

        replayed = False
        if captured is not None and captured.failed is None:
            # the whole body of this micro-batch -- model, loss, backward and,
            # when it closes the step, exchange + optimizer -- replayed,
            # recorded (its first occurrence runs eagerly) or run eagerly
            loss, terms, tags = captured.run(
                batch, (clock.micro - 1) % accumulation_steps, timers)
            replayed = True
        if not replayed:
            loss, terms, tags = process_minibatch(
                model, batch, timers, device, is_raw, evaluator, weights)
            with _timed(timers, 'backprop'):
                if accumulation_steps == 1:
                    unit_backward(loss)     # seed 1.0: no fill, no scaling pass
                else:
                    loss /= accumulation_steps
                    loss.backward()
            if hasattr(model, 'strict'):
                model.strict = False        # layout was validated on batch one

        if not closes_step:
            with _timed(timers, 'logging'):
                sums.add(loss, terms, tags)
        else:
            with _timed(timers, 'optimizer_step'):
                if not replayed:        # a captured / replayed step holds the update
                    if reducer is not None:
                        reducer.wait()
                    optimizer.step()
                    optimizer.zero_grad(set_to_none=True)
            scheduler.step()
            with _timed(timers, 'logging'):
                wanted = clock.step % log_every == 0
                if wanted or accumulation_steps > 1:
                    sums.add(loss, terms, tags)
                if wanted and logger is not None:
                    x = clock.samples
                    sums.write(logger, x, accumulation_steps, _TRAIN_FAMILIES)
                    logger.add_scalar('General/Train loss', sums.loss, x)
                    for g, group in enumerate(optimizer.param_groups):
                        logger.add_scalar(f'General/learning rate/{g}',
                                          group['lr'], x)
                sums.clear()
            for name, hook in hooks.items():
                with _timed(timers, name):
                    hook(clock.step, clock.samples)
            model.train()               # a hook may have switched to eval()

        timers.log(names=list(STAGES) + list(hooks))
        timers('batch_construction').start()
    timers('batch_construction').stop()
    if captured is not None:
        captured.close()


def validate(model, device, loader, samples_passed, logger, evaluator,
             weights=[0.5, 1, 1], is_raw=True):
    """Mean loss and per-scale mean terms over ``loader`` (eval mode,
    no_grad), written against ``samples_passed``."""
    model.eval()
    sums, quiet = ScaleSums(), FakeTimer()
    n_batches = len(loader)
    with torch.no_grad():
        for batch in loader:
            loss, terms, tags = process_minibatch(
                model, batch, quiet, device, is_raw, evaluator, weights)
            sums.add(loss, terms, tags)
    logger.add_scalar('General/Validation loss', sums.loss / n_batches,
                      samples_passed)
    sums.write(logger, samples_passed, n_batches, _VALID_FAMILIES)
