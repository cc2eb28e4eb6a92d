#!/usr/bin/env python3
"""Training entry point with the reference's flags and wiring
(train_flownet.py:31-217): model plugin -> optimizer + LambdaLR ->
``init_losses`` -> ``train``.  Additions: one process per GPU under torchrun
(RCCL gradient all-reduce overlapped with backward) and ``--synthetic``.

The reference's HDF5 data pipeline, TensorBoard writer, serializer and hooks
are outside this build's scope: when this file is
dropped into the reference tree they are imported from there (``utils.*``);
otherwise ``--synthetic`` supplies batches in the same wire format.
"""
import sys
from argparse import ArgumentParser
from pathlib import Path

import torch
import torch.optim as optim

from dvs_of_training_framework_amd import parallel, synthetic
from dvs_of_training_framework_amd.loss import init_losses
from dvs_of_training_framework_amd.model import init_model
from dvs_of_training_framework_amd.optim import FusedAdamW, FusedRAdam, \
    FusedRanger
from dvs_of_training_framework_amd.options import (
    add_train_arguments, add_preprocessed_dataset_arguments,
    validate_train_args)
from dvs_of_training_framework_amd.timer import EventTimer, FakeTimer
from dvs_of_training_framework_amd.training import train

script_dir = Path(__file__).resolve().parent


def parse_args(argv):
    parser = ArgumentParser()
    parser = add_train_arguments(parser)
    parser = add_preprocessed_dataset_arguments(parser)
    args = parser.parse_args(argv)
    args = validate_train_args(args)
    args.model.mkdir(exist_ok=True, parents=True)
    args.log_path = args.model / 'log'
    return args


def get_params2optimize(model):            # train_flownet.py:50-54
    if hasattr(model, 'quantization_layer'):
        return [{'params': model.quantization_layer.parameters()},
                {'params': model.predictor.parameters()}]
    return [{'params': model.parameters()}]


def construct_optimizer(args, params):     # train_flownet.py:57-75
    for g in params:
        g['params'] = list(g['params'])
    params = [g for g in params if g['params']]   # e.g. parameter-free voxeliser
    if args.optimizer == 'ADAM':
        on_gpu = all(p.is_cuda for g in params for p in g['params'])
        opt = FusedAdamW if on_gpu else optim.AdamW
        return opt(params, lr=args.lr, weight_decay=args.wdw, amsgrad=True)
    if args.optimizer in ('RADAM', 'RANGER'):
        # un-vendored submodules upstream (RAdam/, Ranger-Deep-Learning-
        # Optimizer/): fused HIP restatements of the published algorithms
        assert all(p.is_cuda for g in params for p in g['params']), \
            f'--optimizer {args.optimizer} runs on the HIP path only'
        opt = FusedRAdam if args.optimizer == 'RADAM' else FusedRanger
        return opt(params, lr=args.lr, weight_decay=args.wdw)
    assert hasattr(torch.optim, args.optimizer), 'Unknown optimizer type'
    return getattr(torch.optim, args.optimizer)(params, lr=args.lr,
                                                weight_decay=args.wdw)


def make_schedulers(args):                 # train_flownet.py:91-99
    representation_start = args.training_steps * args.rs

    def pred_scheduler(step):
        if step < args.num_warmup_steps:
            return step / args.num_warmup_steps
        return 2 ** (-(step - args.num_warmup_steps) / args.half_life)

    def repr_scheduler(step):
        if step > representation_start:
            return pred_scheduler(step)
        return 0
    return pred_scheduler, repr_scheduler


def construct_train_tools(args, model, passed_steps=0):   # :78-109
    is_splitted = hasattr(model, 'quantization_layer')
    if is_splitted:
        representation_params = [{
            'params': list(model.quantization_layer.parameters()),
            'weight_decay': args.wdw}]
        predictor_params = [{'params': list(model.predictor.parameters())}]
    else:
        representation_params = []
        predictor_params = [{'params': list(model.parameters()),
                             'weight_decay': args.wdw}]
    pred_scheduler, repr_scheduler = make_schedulers(args)
    groups = representation_params + predictor_params
    lambdas = [repr_scheduler] * len(representation_params) + \
        [pred_scheduler] * len(predictor_params)
    keep = [i for i, g in enumerate(groups) if g['params']]
    optimizer = construct_optimizer(args, [groups[i] for i in keep])
    scheduler = optim.lr_scheduler.LambdaLR(
        optimizer, lr_lambda=[lambdas[i] for i in keep])
    for _ in range(passed_steps):
        scheduler.step()
    return optimizer, scheduler


class SyntheticLoader:
    """Endless seeded batches in the reference's wire format
    (utils/dataset.py:961-1020), rank-sharded by seed."""

    def __init__(self, args, rank, steps):
        self.args, self.rank, self.steps = args, rank, steps

    def __iter__(self):
        a = self.args
        seq = a.prefix_length + a.suffix_length + 1
        for i in range(self.steps):
            yield synthetic.to_torch(synthetic.make_batch(
                1234 + self.rank + 1000 * i, a.mbs, a.height, a.width,
                a.synthetic_events, seq_len=seq))


class _Strided:
    """Every rank takes one batch and skips the other ranks' (the loader is
    sequential and cyclic: rank r starts r batches in, set_index above)."""

    def __init__(self, loader, world):
        self.loader, self.world = loader, world

    def __iter__(self):
        return self

    def __next__(self):
        batch = next(self.loader)
        for _ in range(self.world - 1):
            next(self.loader)
        return batch


class _NullLogger:
    def add_scalar(self, *a, **k):
        pass


def main(argv=None):
    args = parse_args(sys.argv[1:] if argv is None else argv)
    device = torch.device(args.device)
    rank, local, world = parallel.init_distributed(device.type)
    if device.type == 'cuda':
        device = torch.device('cuda', local if world > 1 else
                              (device.index or 0))
        torch.cuda.set_device(device)
    timers = EventTimer() if (args.timers and device.type == 'cuda') \
        else FakeTimer()

    model = init_model(args, device)
    parallel.broadcast_parameters(model)
    optimizer, scheduler = construct_train_tools(args, model)
    losses = init_losses(args.shape, args.mbs, model, device,
                         sequence_length=args.prefix_length +
                         args.suffix_length + 1, timers=timers)
    reducer = None
    if world > 1 and hasattr(model, 'predictor'):
        reducer = parallel.GradReducer()
        model.predictor.reducer = reducer

    logger = _NullLogger()
    if rank == 0:
        try:
            from torch.utils.tensorboard import SummaryWriter
            logger = SummaryWriter(str(args.log_path), max_queue=100000000,
                                   flush_secs=100000000)
        except Exception:       # tensorboard is not installed everywhere
            pass

    if args.synthetic:
        loader = SyntheticLoader(args, rank,
                                 args.training_steps * args.accum_step)
    elif getattr(args, 'preprocessed_dataset_path', None) is not None:
        # utils/dataloader.py:89-100: the preprocessed (encoded / quantized)
        # dataset; --compact-events keeps raw events in their 9 B/event columns
        # all the way to the device voxeliser.  Ranks read disjoint strides.
        from dvs_of_training_framework_amd.preprocessed import \
            PreprocessedDataloader
        loader = PreprocessedDataloader(
            path=args.preprocessed_dataset_path, batch_size=args.mbs,
            is_raw=args.is_raw, cache_dir=getattr(args, 'cache_dir', None),
            cache_size=getattr(args, 'cache_size', 0),
            process_only_once=False,
            compact=getattr(args, 'compact_events', False))
        loader.set_index(rank * args.mbs)
        if world > 1:
            loader = _Strided(loader, world)
    else:
        try:    # dropped into the reference tree: use its data pipeline
            from utils.dataloader import get_trainset_params, get_dataloader, \
                choose_data_path
            loader = get_dataloader(get_trainset_params(choose_data_path(args)))
        except ImportError as e:
            raise SystemExit(
                'no dataset pipeline importable (the reference\'s utils.* and '
                f'h5py are needed: {e}); pass --synthetic') from e

    oib = getattr(args, 'optimizer_in_backward', 'auto')
    if device.type == 'cuda' and hasattr(optimizer, 'fuse_into_backward') and \
            hasattr(model, 'predictor') and \
            (oib == 'on' or (oib == 'auto' and reducer is None and
                             getattr(args, 'compute_dtype', 'f32') == 'f32')):
        optimizer.fuse_into_backward(model.predictor)
    if getattr(args, 'device_feeder', False) and device.type == 'cuda' and args.is_raw:
        from dvs_of_training_framework_amd.feed import DeviceFeeder
        loader = DeviceFeeder(loader, device)
    train(model, device, loader, optimizer, args.training_steps,
          scheduler=scheduler, evaluator=losses, logger=logger,
          weights=args.loss_weights, is_raw=args.is_raw,
          accumulation_steps=args.accum_step, timers=timers, hooks={},
          max_events_per_batch=args.max_events_per_batch, reducer=reducer,
          capture=getattr(args, 'capture', False))
    if reducer is not None:
        reducer.close()
    if rank == 0:
        torch.save({'model': model.state_dict(),
                    'optimizer': optimizer.state_dict(),
                    'global_step': args.training_steps},
                   args.model / f'step_{args.training_steps}.pt')


if __name__ == '__main__':
    main()
