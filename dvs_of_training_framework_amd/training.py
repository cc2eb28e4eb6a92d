"""Train / validate loops with the reference's surface (utils/training.py).

``combined_loss`` (:12-24), ``make_hook_periodic`` (:27-30),
``process_minibatch`` (:37-86), ``train`` (:89-235), ``validate`` (:244-271)
keep their signatures, return structures, hook protocol and logging tags.
What changed underneath:
  * the evaluator's fused forward+backward sweep is used when it exists;
  * the 12 lazy ``.item()`` syncs of :77 become ONE packed device->host copy,
    taken only when a term is actually consumed;
  * ``train`` accepts ``reducer`` (parallel.GradReducer): gradient buckets are
    all-reduced on a side stream during backward and joined before
    ``optimizer.step()``; on non-boundary micro-batches the exchange is off.
"""
import torch

from .common import mean
from .loss import unit_backward
from .timer import EventTimer, FakeTimer


def combined_loss(evaluator, flows, flow_ts, flow_sample_idx, images,
                  timestamps, sample_idx, features, weights=[0.5, 1, 1],
                  frame_indices=None):
    kwargs = {} if frame_indices is None else {'frame_indices': frame_indices}
    needs_grad = torch.is_grad_enabled() and any(f.requires_grad
                                                 for f in flows)
    if needs_grad and hasattr(evaluator, 'fused'):
        loss, tt = evaluator.fused(flows, flow_ts, flow_sample_idx, images,
                                   timestamps, sample_idx, weights=weights,
                                   **kwargs)
        return loss, _Terms(tt)
    terms = evaluator(flows, flow_ts, flow_sample_idx, images, timestamps,
                      sample_idx, **kwargs)
    loss = sum(map(lambda v, w: w * mean(v), terms, weights))
    return loss, terms


class _Terms(tuple):
    """((smooth_k), (photo_k), (border_k)) of 0-dim tensors backed by one
    [3,K] tensor, so that host readers can fetch all of it at once."""
    def __new__(cls, packed):
        self = super().__new__(cls, (tuple(r.unbind(0))
                                     for r in packed.unbind(0)))
        self.packed = packed
        return self


def _lazy_items(terms):
    """Generator of generators of floats like utils/training.py:77, but with a
    single device->host copy at first use."""
    cache = []

    def values():
        if not cache:
            if isinstance(terms, _Terms):
                cache.append(terms.packed.detach().cpu().tolist())
            else:
                packed = torch.stack([torch.stack(list(t)) for t in terms])
                cache.append(packed.detach().cpu().tolist())
        return cache[0]
    n_terms, n_scales = len(terms), len(terms[0])
    return ((values()[i][k] for k in range(n_scales))
            for i in range(n_terms))


def make_hook_periodic(hook, checkpointing_interval):
    return lambda step, *args: (None
                                if step % checkpointing_interval
                                else hook(step, *args))


def predictions2tag(predictions):
    return (f'{x.shape[-2]}x{x.shape[-1]}' for x in predictions)


def process_minibatch(model, batch, timers, device, is_raw, evaluator,
                      weights, return_prediction=False):
    timers('batch2gpu').start()
    timestamps, sample_idx, images = map(
        lambda x: x.to(device, non_blocking=True),
        (batch['timestamps'], batch['sample_idx'], batch['images']))
    if is_raw:
        events = batch['events']
        for k in set.difference(set(events.keys()), {'size'}):
            events[k] = events[k].to(device, non_blocking=True)
    else:
        events = batch['data'].to(device, non_blocking=True)
    timers('batch2gpu').stop()
    shape = images.size()[-2:]
    timers('forward').start()
    kwargs = {}
    if hasattr(model, 'last_frame_indices') and 'size' in batch:
        kwargs['batch_size'] = int(batch['size'])   # no device sync for B
    prediction, flow_ts, flow_sample_idx, features = model(
        events, timestamps, sample_idx, shape, raw=is_raw, intermediate=True,
        **kwargs)
    tags = predictions2tag(prediction)
    timers('forward').stop()
    timers('loss').start()
    loss, terms = combined_loss(
        evaluator, prediction, flow_ts, flow_sample_idx, images, timestamps,
        sample_idx, features, weights=weights,
        frame_indices=getattr(model, 'last_frame_indices', None))
    terms = _lazy_items(terms)
    timers('loss').stop()
    add_info = tuple()
    if return_prediction:
        add_info = ({'prediction': prediction, 'flow_ts': flow_ts,
                     'flow_sample_idx': flow_sample_idx,
                     'features': features}, )
    return (loss, terms, tags) + add_info


def train(model, device, loader, optimizer, num_steps: int, scheduler, logger,
          evaluator, weights=[0.5, 1, 1], is_raw=True, accumulation_steps=1,
          timers=None, hooks={}, init_step=0, init_samples_passed=0,
          max_events_per_batch: int = 350000, reducer=None,
          log_every: int = 1):
    """Reference semantics (utils/training.py:89-235): micro-batch
    accumulation, oversize-batch skip, optimizer/scheduler step on the
    boundary, per-scale TensorBoard scalars against samples_passed, hooks
    ``Callable(step, samples_passed)`` after every optimizer step.

    reducer:   parallel.GradReducer for data-parallel training
    log_every: write scalars every n-th optimizer step (1 = the reference)
    """
    if timers is None:
        timers = EventTimer() if torch.device(device).type == 'cuda' \
            else FakeTimer()
    model.train()
    samples_passed = init_samples_passed
    loss_sum = 0
    smooth_sum, photo_sum, out_reg_sum = [], [], []
    optimizer.zero_grad(set_to_none=True)
    init_batch = init_step * accumulation_steps
    global_step = init_batch
    num_skipped = 0
    timers('batch_construction').start()
    for batch in loader:
        if global_step == num_steps * accumulation_steps:
            break
        num_events = batch['events']['x'].numel() if is_raw else 0
        if num_events > max_events_per_batch:
            num_skipped += 1
            num_processed = global_step - init_batch
            print(f'Skipping batch with {num_events} events')
            print('Augmentation parameters '
                  f'{batch["augmentation_params"]}')
            print('Processing rate is '
                  f'{num_processed / (num_processed + num_skipped):.2f}')
            continue
        global_step += 1
        timers('batch_construction').stop()
        samples_passed += batch['size']
        is_step_boundary = global_step % accumulation_steps == 0
        if reducer is not None:
            reducer.enabled = is_step_boundary
        if hasattr(optimizer, 'fused_active'):    # optim.fuse_into_backward
            optimizer.fused_active = is_step_boundary
        loss, (smoothness, photometric, out_reg), tags = process_minibatch(
            model, batch, timers, device, is_raw, evaluator, weights)
        timers('backprop').start()
        if accumulation_steps == 1:
            unit_backward(loss)         # seed 1.0 without a fill / scaling pass
        else:
            loss /= accumulation_steps
            loss.backward()
        timers('backprop').stop()
        if hasattr(model, 'strict'):
            model.strict = False    # layout was validated on the first batch

        do_log = is_step_boundary and \
            (global_step // accumulation_steps) % log_every == 0
        if is_step_boundary:
            timers('optimizer_step').start()
            if reducer is not None:
                reducer.wait()
            optimizer.step()
            optimizer.zero_grad(set_to_none=True)
            timers('optimizer_step').stop()
            scheduler.step()

            timers('logging').start()
            if do_log or accumulation_steps > 1:
                photo_sum = add_loss(photo_sum, photometric)
                smooth_sum = add_loss(smooth_sum, smoothness)
                out_reg_sum = add_loss(out_reg_sum, out_reg)
                loss_sum += loss.item()
            if do_log and logger is not None:
                for tag, s, p, o in zip(tags, smooth_sum, photo_sum,
                                        out_reg_sum):
                    logger.add_scalar(f'Train/photometric loss/{tag}',
                                      p / accumulation_steps, samples_passed)
                    logger.add_scalar(f'Train/smoothness loss/{tag}',
                                      s / accumulation_steps, samples_passed)
                    logger.add_scalar(f'Train/out regularization/{tag}',
                                      o / accumulation_steps, samples_passed)
                logger.add_scalar('General/Train loss', loss_sum,
                                  samples_passed)
                for i, lr in enumerate([p['lr']
                                        for p in optimizer.param_groups]):
                    logger.add_scalar(f'General/learning rate/{i}', lr,
                                      samples_passed)
            loss_sum = 0
            smooth_sum, photo_sum, out_reg_sum = [], [], []
            timers('logging').stop()

            step = global_step // accumulation_steps
            for k, hook in hooks.items():
                timers(k).start()
                hook(step, samples_passed)
                timers(k).stop()
            # make sure to return to train after all hooks
            model.train()
        else:
            timers('logging').start()
            photo_sum = add_loss(photo_sum, photometric)
            smooth_sum = add_loss(smooth_sum, smoothness)
            out_reg_sum = add_loss(out_reg_sum, out_reg)
            loss_sum += loss.item()
            timers('logging').stop()

        timers.log(names=['batch_construction', 'batch2gpu', 'forward',
                          'loss', 'backprop', 'optimizer_step', 'logging'] +
                   list(hooks))
        timers('batch_construction').start()
    timers('batch_construction').stop()


def add_loss(loss_sum, loss_values):
    if len(loss_sum) == 0:
        return list(loss_values)
    return [x + y for x, y in zip(loss_sum, loss_values)]


def validate(model, device, loader, samples_passed, logger, evaluator,
             weights=[0.5, 1, 1], is_raw=True):
    model.eval()
    n = len(loader)
    photo_sum, smooth_sum, out_reg_sum = [], [], []
    loss_sum = 0
    with torch.no_grad():
        for batch in loader:
            loss, (smoothness, photometric, out_reg), tags = \
                process_minibatch(model, batch, FakeTimer(), device, is_raw,
                                  evaluator, weights)
            photo_sum = add_loss(photo_sum, photometric)
            smooth_sum = add_loss(smooth_sum, smoothness)
            out_reg_sum = add_loss(out_reg_sum, out_reg)
            loss_sum += loss.item()
    logger.add_scalar('General/Validation loss', loss_sum / n, samples_passed)
    for tag, s, p, o in zip(tags, smooth_sum, photo_sum, out_reg_sum):
        logger.add_scalar(f'Validation/smoothness loss/{tag}', s / n,
                          samples_passed)
        logger.add_scalar(f'Validation/photometric loss/{tag}', p / n,
                          samples_passed)
        logger.add_scalar(f'Validation/out regularization loss/{tag}', o / n,
                          samples_passed)
