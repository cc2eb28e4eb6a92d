"""Command-line flags of the training entry point, restated 1:1 from the
reference (utils/options.py:10-347) so existing launch scripts keep working;
``--world-size`` / ``--synthetic*`` are additions for the MI355X build.
The reference module itself cannot be imported here (it imports the absent
``mish`` submodule at :7)."""
import os
from pathlib import Path

import torch
import torch.nn as nn


class Mish(nn.Module):
    """x * tanh(softplus(x)) (Misra 2019) -- stands in for the un-vendored
    ``mish.mish.Mish`` (utils/options.py:7); the HIP conv stack fuses it."""

    def forward(self, x):
        return nn.functional.mish(x)


def add_common_arguments(parser):          # utils/options.py:10-19
    parser.add_argument('--allow-obsolete-code', action='store_true')
    parser.add_argument('--allow-arguments-change', action='store_true')
    return parser


def add_model_arguments(parser):           # utils/options.py:22-56
    parser.add_argument('--flownet_path', default=Path('EV_FlowNet'),
                        type=Path, required=False,
                        help='relative path to a model to train')
    parser.add_argument('--mish', action='store_true')
    parser.add_argument('-d', '--device', default=torch.device('cuda:0'),
                        type=torch.device, required=False)
    parser.add_argument('-bs', '--batch_size', dest='bs', default=32,
                        type=int, required=False,
                        help='batch size for an optimizer step')
    parser.add_argument('--profiling', choices=['CPU', 'NVTX', 'None'],
                        default='None')
    parser.add_argument('-sp', '--starting_point', dest='sp', default=None,
                        required=False)
    return parser


def add_dataset_arguments(parser):         # utils/options.py:59-113
    parser.add_argument('--ev_images', action='store_true')
    parser.add_argument('-cl', '--collapse_length', dest='cl', default=6,
                        type=int, required=False)
    parser.add_argument('--height', dest='height', default=256, type=int)
    parser.add_argument('--width', dest='width', default=256, type=int)
    parser.add_argument('--min-sequence-length', dest='min_sequence_length',
                        default=1, type=int)
    parser.add_argument('--max-sequence-length', dest='max_sequence_length',
                        default=1, type=int)
    parser.add_argument('--prefix-length', dest='prefix_length', default=0,
                        type=int)
    parser.add_argument('--suffix-length', dest='suffix_length', default=0,
                        type=int)
    parser.add_argument('--dynamic-sample-length',
                        dest='dynamic_sample_length', action='store_true')
    parser.add_argument('--event-representation-depth',
                        dest='event_representation_depth', default=9,
                        type=int)
    return parser


def add_dataloader_arguments(parser):      # utils/options.py:116-129
    parser.add_argument('-mbs', '--micro_batch_size', dest='mbs', default=32,
                        type=int, required=False,
                        help='batch size for a single forward-backward pass')
    parser.add_argument('--num_workers', dest='num_workers',
                        default=len(os.sched_getaffinity(0)), type=int)
    return parser


def add_preprocessed_dataset_arguments(parser):   # utils/options.py:150-173
    parser.add_argument('--preprocessed-dataset-path',
                        dest='preprocessed_dataset_path', default=None,
                        type=Path)
    parser.add_argument('--cache-dir', dest='cache_dir', default=None,
                        type=Path)
    parser.add_argument('--cache-size', dest='cache_size', default=5,
                        type=int)
    parser.add_argument('--process-only-once', dest='process_only_once',
                        action='store_true')
    return parser


def add_train_arguments(parser):           # utils/options.py:204-302
    parser = add_common_arguments(parser)
    parser = add_model_arguments(parser)
    parser = add_dataset_arguments(parser)
    parser = add_dataloader_arguments(parser)
    parser.add_argument('-m', '--model', required=True, type=Path,
                        help='Directory to store learned weights')
    parser.add_argument('--half_life', dest='half_life', default=100000,
                        type=float)
    parser.add_argument('-wdw', '--weight_decay_weight', dest='wdw',
                        default=1e-4, type=float)
    parser.add_argument('-ne', '--num_training_steps', dest='training_steps',
                        default=1000000, type=int)
    parser.add_argument('--num-warmup-steps', dest='num_warmup_steps',
                        default=0, type=int)
    parser.add_argument('-lr', '--learning_rate', dest='lr', default=1e-3,
                        type=float)
    parser.add_argument('-vp', '--validation_period', dest='vp', default=1000,
                        type=int)
    parser.add_argument('--optimizer', default='RANGER',
                        choices=['ADAM', 'RADAM', 'RANGER'])
    parser.add_argument('--loss_weights', default=[0.5, 1, 1], nargs=3,
                        type=float)
    parser.add_argument('--representation-start', dest='rs', default=0.5,
                        type=float)
    parser.add_argument('--num_checkpoints', dest='num_checkpoints',
                        default=2, type=int)
    parser.add_argument('--permanent_interval', dest='permanent_interval',
                        default=10000, type=int)
    parser.add_argument('--checkpointing_interval',
                        dest='checkpointing_interval', default=1000, type=int)
    parser.add_argument('--timers', dest='timers', action='store_true')
    parser.add_argument('--do_not_continue', dest='do_not_continue',
                        action='store_true')
    parser.add_argument('--max-events-per-batch', dest='max_events_per_batch',
                        default=35000000, type=int)
    parser.add_argument('--skip-validation', dest='skip_validation',
                        action='store_true')
    # --- additions of this build
    parser.add_argument('--compute-dtype', dest='compute_dtype', default='f32',
                        choices=['f32', 'bf16x3', 'bf16', 'bf16s'],
                        help='matrix-core operand type of the conv stack: exact f32, '
                             'bf16 hi+lo split (three products, ~f32 accuracy), bf16 '
                             '(operands rounded in registers; storage and accumulation '
                             'f32) or bf16s (bf16 twins of activations / gradients / '
                             'weight forms streamed through LDS; master weights, '
                             'gradients of weights and optimizer state stay f32)')
    parser.add_argument('--capture', action='store_true',
                        help='replay the loop body from one C call per micro-batch '
                             'once a batch signature has been seen (capture.CapturedLoop '
                             '/ the step executor; ADAM; with or without gradient '
                             'accumulation, with or without data parallelism -- the '
                             'executor then issues the gradient exchange)')
    parser.add_argument('--optimizer-in-backward', dest='optimizer_in_backward',
                        default='auto', choices=['auto', 'on', 'off'],
                        help='update a gradient bucket\'s parameters as soon as its '
                             'gradients are final (behind its all-reduce under data '
                             'parallelism) instead of in optimizer.step() '
                             '(optim.fuse_into_backward; same arithmetic).  auto: on '
                             'for --compute-dtype f32 in a single process, where the '
                             'HBM-bound update hides beside matrix-bound kernels '
                             '(+1.5 %%); off for the bf16 modes, which are '
                             'bandwidth-bound themselves, and under data parallelism '
                             '(the per-bucket waits for the exchange cost more than '
                             'the overlap gives: measured in a 1-rank group)')
    parser.add_argument('--device-feeder', dest='device_feeder', action='store_true',
                        help='move batches to the device on a copy stream, one step '
                             'ahead (feed.DeviceFeeder), instead of tensor.to(device) '
                             'at the top of every step (utils/training.py:45-56)')
    parser.add_argument('--compact-events', dest='compact_events',
                        action='store_true',
                        help='with --preprocessed-dataset-path: hand raw '
                             'events to the device voxeliser in their 9 B/event '
                             'encoded columns (no int64 wire columns)')
    parser.add_argument('--synthetic', action='store_true',
                        help='train on seeded synthetic batches (no dataset)')
    parser.add_argument('--synthetic-events', dest='synthetic_events',
                        default=None, type=int,
                        help='events per synthetic sample (default H*W)')
    return parser


def validate_dataset_args(args):           # utils/options.py:305-309
    args.is_raw = not args.ev_images
    args.shape = (args.height, args.width)
    assert args.prefix_length + args.suffix_length < args.max_sequence_length
    return args


def validate_train_args(args):             # utils/options.py:318-325
    args = validate_dataset_args(args)
    assert args.bs > 0
    assert args.mbs > 0
    assert args.bs % args.mbs == 0
    args.accum_step = args.bs // args.mbs
    assert args.permanent_interval % args.checkpointing_interval == 0
    return args


def options2dataset_kwargs(parameters):    # utils/options.py:332-338
    return dict(prefix_length=parameters.prefix_length,
                suffix_length=parameters.suffix_length,
                max_sequence_length=parameters.max_sequence_length,
                dynamic_sample_length=parameters.dynamic_sample_length,
                event_representation_depth=parameters.
                event_representation_depth)


def options2model_kwargs(parameters):      # utils/options.py:341-347
    kwargs = options2dataset_kwargs(parameters)
    kwargs['activation'] = Mish() if parameters.mish else nn.ReLU()
    # MI355X build only; models without this ctor argument never see it
    # (model.init_model filters by signature, utils/model.py:10-23)
    if getattr(parameters, 'compute_dtype', 'f32') != 'f32':
        kwargs['compute_dtype'] = parameters.compute_dtype
    return kwargs
