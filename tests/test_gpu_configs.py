"""BASELINE.json configurations exercised AS CONFIGURED on the GPU
(VERDICT round 1, item 1):

  configs[2]  256x256x5, 8 samples per GPU, bf16 matrix-core operands
  configs[4]  Ranger + bf16 + 512x512x12 + 1 M events per sample, streamed
              through the 9 B/event compact columns
  DP claim    gradient of a batch == mean of the gradients of its halves
              (normalisers of utils/loss.py:101-113)
  RCCL        the bucketed all-reduce path on hardware, in a child process
              with a 1-rank process group, bit-identical to the plain run
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from dvs_of_training_framework_amd import synthetic
from oracle import cpu_oracle as orc
from oracle.ref_model import ref_predictor

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = Path(__file__).resolve().parent.parent


# ------------------------------------------------------------------ configs[2]
@pytest.fixture(scope='module')
def cfg2_case():
    """One predictor (seeded), one input, float64 reference outputs and
    parameter gradients at B=8, 256x256x5 -- shared by the operand modes."""
    from dvs_of_training_framework_amd.predictor import Predictor
    torch.manual_seed(31)
    B, C, H, W = 8, 5, 256, 256
    net = Predictor(C).cuda()
    x = torch.randn(B, C, H, W, device=DEV)
    seeds = [torch.randn(B, 2, H // s, W // s, device=DEV) * (0.5 / s) for s in (8, 4, 2, 1)]
    state = {k: v.detach().cpu().double().contiguous().requires_grad_(True)
             for k, v in net.state_dict().items()}
    ref = ref_predictor(state, x.cpu().double())
    torch.autograd.backward(ref, [g.cpu().double() for g in seeds])
    return dict(net=net, x=x, seeds=seeds,
                flows=[r.detach().float() for r in ref],
                grads={k: v.grad.float() for k, v in state.items()})


@pytest.mark.parametrize('dtype,flow_tol,grad_rel,grad_cos', [
    # any fp32-accurate path differs from float64 by ReLU-mask flips of
    # activations within ~1e-6 of zero: measured per-tensor gradient distance
    # of ATen/MIOpen's own GPU fp32 path at this size 0.3e-3..4.7e-3
    # (tools/dbg_fullsize.py); this mode 0.3e-3..6.3e-3 (largest on the 512-channel
    # residual layers, which run as Winograd F(2x2) in this mode)
    ('bf16x3', 1e-3, 1e-2, 0.9999),
    ('bf16', 3e-2, 0.2, 0.98),          # the bf16 bars of test_gpu_model.py:289-311
    # bf16 twins: activations / gradients / prepared weights ALSO stored as bf16
    # and streamed through LDS in that form by the forward / data-gradient
    # kernels (one more rounding per tensor than 'bf16': the stored twin)
    ('bf16s', 3e-2, 0.25, 0.97),
])
def test_config2_workload_in_bf16_modes_vs_float64(cfg2_case, dtype, flow_tol, grad_rel,
                                                  grad_cos):
    from dvs_of_training_framework_amd.predictor import Predictor
    c = cfg2_case
    net = Predictor(5, compute_dtype=dtype).cuda()
    net.load_state_dict(c['net'].state_dict())
    flows = net(c['x'])
    torch.autograd.backward(flows, c['seeds'])
    torch.cuda.synchronize()
    for f, r in zip(flows, c['flows']):
        r = r.cuda()
        assert float((f.detach() - r).norm()) <= flow_tol * float(r.norm())
    for name, p in net.named_parameters():
        r = c['grads'][name].cuda()
        g = p.grad
        assert bool(torch.isfinite(g).all()), name
        rel = float((g - r).norm()) / (float(r.norm()) + 1e-12)
        cos = float((g * r).sum()) / (float(g.norm() * r.norm()) + 1e-12)
        assert rel <= grad_rel and cos >= grad_cos, (name, rel, cos)


# ------------------------------------------------------------------ configs[4]
def test_config4_step_ranger_bf16_one_million_events_compact():
    """One whole training step as configs[4] names it: Ranger, bf16 operands,
    512x512x12, 1 M events per sample fed as the reference's encoded columns
    (9 B/event) -- against the CPU port at batch 2: voxel grid (indices exact,
    sums 1e-3), flows / loss at bf16 accuracy, gradient direction, and the
    Ranger update itself at f32 accuracy (restatement fed the SAME gradients)."""
    from dvs_of_training_framework_amd import encoding
    from dvs_of_training_framework_amd.loss import init_losses
    from dvs_of_training_framework_amd.net import Model
    from dvs_of_training_framework_amd.optim import FusedRanger
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch
    from oracle.ref_optim import RefRanger
    B, C, H, W, n = 2, 12, 512, 512, 1_000_000
    b_np = synthetic.make_batch(41, B, H, W, n)
    wire = synthetic.to_torch(b_np)
    enc = encoding.encode_batch(wire['events'], wire['timestamps'], wire['sample_idx'],
                                wire['images'], {}, B)
    assert enc['events']['x'].dtype == torch.short and enc['events']['polarity'].dtype == torch.bool
    batch = {'events': encoding.compact_events(enc), 'timestamps': wire['timestamps'],
             'sample_idx': wire['sample_idx'], 'images': wire['images'], 'size': B}

    torch.manual_seed(7)
    model = Model(DEV, event_representation_depth=C, compute_dtype='bf16')
    model.train()
    opt = FusedRanger(model.predictor.parameters(), lr=1e-3, weight_decay=1e-4)
    ev = init_losses((H, W), B, model, DEV, sequence_length=1)
    before = {k: v.detach().cpu().clone() for k, v in model.predictor.state_dict().items()}

    # voxeliser alone first (debug outputs): integer parts bit-exact
    from dvs_of_training_framework_amd import voxel
    dev_ev = {k: v.to(DEV) for k, v in batch['events'].items()}
    t0 = torch.zeros(B, device=DEV)
    t1 = torch.full((B,), synthetic.WINDOW, device=DEV)
    grid, bin0, lin0 = voxel.voxelize_compact(dev_ev, t0, t1, B, C, H, W, debug=True)
    want, obin, olin = orc.voxelize(b_np['events'], np.zeros(B, np.float32),
                                    np.full(B, synthetic.WINDOW, np.float32), B, C, H, W)
    assert np.array_equal(bin0.cpu().numpy(), obin)
    assert np.array_equal(lin0.cpu().numpy(), olin)
    np.testing.assert_allclose(grid.cpu().numpy(), want, atol=1e-4, rtol=1e-3)

    loss, terms, tags, info = process_minibatch(model, batch, FakeTimer(), DEV, True, ev,
                                                [0.5, 1, 1], return_prediction=True)
    loss.backward()
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.predictor.named_parameters()}
    opt.step()
    torch.cuda.synchronize()

    # CPU port, f32
    state = {k: v.clone().requires_grad_(True) for k, v in before.items()}
    flows = ref_predictor(state, torch.from_numpy(want))
    o_terms, o_loss, o_grads = orc.losses(
        [f.detach().numpy() for f in flows], b_np['timestamps'].reshape(B, 2), np.arange(B),
        b_np['images'], b_np['timestamps'], b_np['sample_idx'])
    torch.autograd.backward(flows, [torch.from_numpy(g) for g in o_grads])
    for f, r in zip(info['prediction'], flows):
        r = r.detach()
        assert float((f.detach().cpu() - r).norm()) <= 3e-2 * float(r.norm())
    assert abs(float(loss.detach()) - o_loss) <= 2e-2 * abs(o_loss)
    for k, g in grads.items():
        r = state[k].grad
        cos = float((g * r).sum()) / (float(g.norm() * r.norm()) + 1e-12)
        assert bool(torch.isfinite(g).all()) and cos >= 0.95, (k, cos)
    # the optimizer: restatement stepped with the GPU's own gradients
    names = list(before)
    ref_params = [before[k].clone().requires_grad_(True) for k in names]
    ro = RefRanger(ref_params, lr=1e-3, weight_decay=1e-4)
    for k, p in zip(names, ref_params):
        p.grad = grads[k]
    ro.step()
    after = model.predictor.state_dict()
    for k, p in zip(names, ref_params):
        err = (after[k].cpu() - p.detach()).abs().max()
        assert err <= 5e-6 * p.detach().abs().max() + 1e-7, (k, float(err))
    assert list(tags) == ['64x64', '128x128', '256x256', '512x512']


# ------------------------------------------------- whole steps, as configured
def _whole_step_vs_cpu_port(B, C, H, W, tags_want, dtype='f32', seed=3, batch_seed=1234,
                            flow_tol=1e-3, term_rtol=1e-3, loss_rtol=1e-3, grad_rel=1e-2,
                            grad_cos=None):
    """ONE training step -- voxelise -> predictor -> multi-scale loss -> backward
    -> AdamW -- on the GPU against the CPU port (oracle/: C voxeliser + loss,
    ATen f32 predictor) at the SAME batch: voxel bin / linear indices bit-exact,
    grid sums 1e-4, flows `flow_tol` of the peak per scale, the twelve loss
    terms and the loss, parameter gradients against ATen autograd on the field
    norm (`grad_rel`; `grad_cos` instead for the bf16 storage mode), and the
    update moves the weights."""
    from dvs_of_training_framework_amd import voxel
    from dvs_of_training_framework_amd.loss import init_losses
    from dvs_of_training_framework_amd.net import Model
    from dvs_of_training_framework_amd.optim import FusedAdamW
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch
    n = H * W
    b_np = synthetic.make_batch(batch_seed, B, H, W, n)
    batch = synthetic.to_torch(b_np, DEV)
    torch.manual_seed(seed)
    model = Model(DEV, event_representation_depth=C, compute_dtype=dtype)
    model.train()
    opt = FusedAdamW(model.predictor.parameters(), lr=1e-3, weight_decay=1e-4, amsgrad=True)
    ev = init_losses((H, W), B, model, DEV, sequence_length=1)
    before = {k: v.detach().cpu().clone() for k, v in model.predictor.state_dict().items()}

    # voxeliser: integer parts bit-exact against the C oracle
    t0 = torch.zeros(B, device=DEV)
    t1 = torch.full((B,), synthetic.WINDOW, device=DEV)
    grid, bin0, lin0 = voxel.voxelize(batch['events'], t0, t1, B, C, H, W, debug=True)
    want, obin, olin = orc.voxelize(b_np['events'], np.zeros(B, np.float32),
                                    np.full(B, synthetic.WINDOW, np.float32), B, C, H, W)
    assert np.array_equal(bin0.cpu().numpy(), obin)
    assert np.array_equal(lin0.cpu().numpy(), olin)
    np.testing.assert_allclose(grid.cpu().numpy(), want, atol=1e-4, rtol=1e-4)
    del grid, bin0, lin0

    loss, terms, tags, info = process_minibatch(model, batch, FakeTimer(), DEV, True, ev,
                                                [0.5, 1, 1], return_prediction=True)
    assert list(tags) == tags_want
    loss.backward()
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.predictor.named_parameters()}
    got_terms = np.array(terms.host())
    opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(before['enc.0.conv.weight'],
                           model.predictor.state_dict()['enc.0.conv.weight'].cpu())

    # the CPU port of the same step
    state = {k: v.clone().requires_grad_(True) for k, v in before.items()}
    flows = ref_predictor(state, torch.from_numpy(want))
    o_terms, o_loss, o_grads = orc.losses(
        [f.detach().numpy() for f in flows], b_np['timestamps'].reshape(B, 2), np.arange(B),
        b_np['images'], b_np['timestamps'], b_np['sample_idx'])
    for f, r in zip(info['prediction'], flows):
        r = r.detach()
        assert float((f.detach().cpu() - r).abs().max()) <= flow_tol * float(r.abs().max())
    np.testing.assert_allclose(got_terms, o_terms, rtol=term_rtol, atol=1e-7)
    assert abs(float(loss.detach()) - o_loss) <= loss_rtol * abs(o_loss)
    torch.autograd.backward(flows, [torch.from_numpy(g) for g in o_grads])
    for k, g in grads.items():
        r = state[k].grad
        assert bool(torch.isfinite(g).all()), k
        if grad_cos is not None:
            cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-30))
            assert cos >= grad_cos, (k, cos)
            continue
        # field norm: see test_predictor_full_size_vs_float64_and_determinism (ReLU-mask
        # flips of activations within 1e-6 of zero) and DESIGN section 2 (loss gradients
        # where |warped - prev| < 1e-3)
        assert float((g - r).norm()) <= grad_rel * float(r.norm()) + 1e-9, \
            (k, float((g - r).norm()) / float(r.norm()))


def test_config1_step_as_benchmarked_vs_the_cpu_port():
    """BASELINE.json configs[1], the configuration bench.py's headline number
    is quoted on: EV_FlowNet 256x256x5, batch 8, 65 536 events per sample
    (bench.py's own seeded batch), exact f32 -- one whole step against the CPU
    port (round-3 verdict: the pieces were tested apart, the step was not)."""
    _whole_step_vs_cpu_port(8, 5, 256, 256, ['32x32', '64x64', '128x128', '256x256'])


def test_config2_bf16_twins_step_vs_the_cpu_port():
    """configs[2] per GPU in the bf16 twins mode (activations / gradients /
    prepared weights streamed as bf16, f32 accumulation, f32 loss and
    optimizer): the same whole step against the exact-f32 CPU port at bf16
    tolerances -- flows 3e-2 of the peak per scale (the bar of the per-mode
    predictor test), loss terms 2e-2, parameter gradients cosine >= 0.97.
    Voxel indices stay bit-exact (the voxeliser does not change with the mode)."""
    _whole_step_vs_cpu_port(8, 5, 256, 256, ['32x32', '64x64', '128x128', '256x256'],
                            dtype='bf16s', flow_tol=3e-2, term_rtol=2e-2, loss_rtol=2e-2,
                            grad_cos=0.97)


# ------------------------------------------------------------------ configs[3]
def test_config3_step_as_configured_vs_the_cpu_port():
    """BASELINE.json configs[3] per GPU, as configured: EV_FlowNet on
    480x640x9-bin (MVSEC-shaped) frames, batch 16 over 4 GPUs = 4 per GPU,
    H*W = 307 200 events per sample, multi-scale warp loss, f32."""
    _whole_step_vs_cpu_port(4, 9, 480, 640, ['60x80', '120x160', '240x320', '480x640'])


# ----------------------------------------------------------- DP equivalence
def _split(b_np, B, lo, hi):
    ev = b_np['events']
    m = (ev['sample_index'] >= lo) & (ev['sample_index'] < hi)
    events = {k: v[m].copy() for k, v in ev.items()}
    events['sample_index'] = events['sample_index'] - lo
    return {'events': events, 'timestamps': b_np['timestamps'][2 * lo:2 * hi].copy(),
            'sample_idx': b_np['sample_idx'][2 * lo:2 * hi] - lo,
            'images': b_np['images'][2 * lo:2 * hi].copy(), 'augmentation_params': {},
            'size': hi - lo}


def test_gradient_of_a_batch_is_the_mean_of_its_halves():
    """What the gradient all-reduce (average over ranks of equal per-rank
    batch) relies on: with the loss normalised per sample by LOCAL N
    (1/N means, 2*c_n*N for the border term, utils/loss.py:101-113) the
    global-batch gradient equals the mean of the shard gradients."""
    from dvs_of_training_framework_amd.loss import init_losses
    from dvs_of_training_framework_amd.net import Model
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch
    B, H, W, C = 8, 128, 128, 5
    torch.manual_seed(13)
    model = Model(DEV, event_representation_depth=C)
    model.train()
    b_np = synthetic.make_batch(23, B, H, W, 20000)

    def grads(batch_np):
        n = batch_np['size']
        ev = init_losses((H, W), n, model, DEV, sequence_length=1)
        for p in model.parameters():
            p.grad = None
        model.strict = True
        loss, _, _ = process_minibatch(model, synthetic.to_torch(batch_np, DEV), FakeTimer(),
                                       DEV, True, ev, [0.5, 1, 1])
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), [p.grad.detach().clone() for p in model.parameters()]
    l_all, g_all = grads(b_np)
    l_a, g_a = grads(_split(b_np, B, 0, 4))
    l_b, g_b = grads(_split(b_np, B, 4, 8))
    assert abs(l_all - 0.5 * (l_a + l_b)) <= 1e-5 * abs(l_all)
    # The identity is exact in real arithmetic.  In f32 the batch-8 and batch-4
    # runs take different kernels (the residual layers reach the Winograd gate
    # at 8 samples only; K splits of the weight gradients depend on the batch),
    # whose ~1e-6 activation differences flip ReLU masks here and there:
    # measured 5e-4 on the first encoder layer (the end of the backward
    # chain), 1e-5..1e-4 elsewhere.  A wrong normaliser would show as O(1).
    worst = {}
    for (name, _), g, a, b in zip(model.named_parameters(), g_all, g_a, g_b):
        mean = 0.5 * (a + b)
        worst[name] = float((g - mean).norm()) / (float(g.norm()) + 1e-12)
    assert max(worst.values()) <= 2e-3, worst
    assert sorted(worst.values())[len(worst) // 2] <= 3e-4, worst


# ------------------------------------------------------------------- RCCL
def test_rccl_bucket_path_is_bit_identical_in_a_one_rank_group():
    """DVSOF_FORCE_DIST=1 in a FRESH process: a 1-rank nccl (= RCCL) group,
    buckets all-reduced on the exchange stream as the backward completes
    them.  Gradients must equal the no-reducer single-stream run bit for bit
    (AVG over one rank is the identity), also with the optimizer fused into
    the bucket hooks (one update per bucket behind its collective; weights
    compared after the step) -- the cross-stream ordering of buckets whose
    units are enqueued on both backward streams (enc.2 | enc.1, enc.0)."""
    env = dict(os.environ, DVSOF_FORCE_DIST='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(29500 + os.getpid() % 2000), RANK='0', WORLD_SIZE='1',
               LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, str(ROOT / 'tests' / 'dp_child.py')], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res['backend'] == 'nccl' and res['world'] == 1
    assert res['buckets_reduced'] == 8 * res['reduced_runs']
    assert res['bytes_reduced'] == res['grad_bytes'] * res['reduced_runs']
    assert res['grads_bit_identical'], res
    assert res['fused_weights_bit_identical'], res
    assert res['repeat_bit_identical'], res
    # dvsof_allreduce_bucket: the C ABI's own RCCL communicator
    assert res['direct_bit_identical'], res
    assert res['direct_bytes'] == res['grad_bytes'] * 4
