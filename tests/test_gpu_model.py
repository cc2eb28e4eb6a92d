"""End-to-end parity on the GPU: Model plugin contract, whole training step
against the CPU port (oracle/), fused AdamW against torch.optim.AdamW,
train loop, OpticalFlow wrapper."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from dvs_of_training_framework_amd import synthetic
from oracle import cpu_oracle as orc
from oracle.ref_model import ref_predictor

pytestmark = pytest.mark.gpu
GOLD = json.loads((Path(__file__).parent / 'golden' / 'plumbing.json').read_text())
DEV = 'cuda'


def make_model(depth=3, **kw):
    from dvs_of_training_framework_amd.net import Model
    torch.manual_seed(0)
    return Model(torch.device(DEV), event_representation_depth=depth, **kw)


def test_forward_contract_matches_dummynet_witness():
    # config 1 shapes (DummyNet, B=4, 64x64): SURVEY App. B / plumbing.json
    model = make_model(3)
    batch = synthetic.to_torch(synthetic.make_batch(1234, 4, 64, 64), DEV)
    with torch.no_grad():
        flows, flow_ts, fsi, feats = model(batch['events'], batch['timestamps'],
                                           batch['sample_idx'], (64, 64), raw=True,
                                           intermediate=True)
    g = GOLD['cfg1']
    assert [list(f.shape) for f in flows] == g['shapes']
    np.testing.assert_allclose(flow_ts.cpu().numpy(), g['flow_ts'], rtol=1e-6)
    assert fsi.tolist() == g['flow_sample_idx'] and feats == ()
    assert all(f.dtype == torch.float32 for f in flows)
    assert hasattr(model, 'quantization_layer') and hasattr(model, 'predictor')
    assert 'predictor.enc.0.conv.bias' in model.state_dict()


def test_init_losses_probes_with_zero_events():
    from dvs_of_training_framework_amd.loss import init_losses
    ev = init_losses((32, 48), 2, make_model(5), DEV, sequence_length=1)
    assert ev.shapes == [(4, 6), (8, 12), (16, 24), (32, 48)]


def test_quantize_and_padding_paths():
    model = make_model(5)
    b_np = synthetic.make_batch(3, 2, 40, 56, 3000)       # not multiples of 16
    b = synthetic.to_torch(b_np, DEV)
    grid = model.quantize(b['events'], b['timestamps'], b['sample_idx'], (40, 56))
    want, _, _ = orc.voxelize(b_np['events'], np.zeros(2, np.float32),
                              np.full(2, 0.04, np.float32), 2, 5, 40, 56)
    assert grid.shape == (2, 5, 40, 56)
    np.testing.assert_allclose(grid.cpu().numpy(), want, atol=2e-5)
    with torch.no_grad():
        flows, _, _ = model(b['events'], b['timestamps'], b['sample_idx'], (40, 56))
    assert [tuple(f.shape[2:]) for f in flows] == [(5, 7), (10, 14), (20, 28), (40, 56)]
    # quantized input (raw=False) takes the same predictor path
    with torch.no_grad():
        flows2, _, _ = model(grid, b['timestamps'], b['sample_idx'], (40, 56), raw=False)
    for a, c in zip(flows, flows2):
        assert torch.allclose(a, c, atol=1e-6)


@pytest.mark.parametrize('mish', [False, True])
def test_training_step_vs_cpu_port(mish):
    """voxelise -> predictor -> loss -> backward, HIP vs the CPU port: loss and
    flows within 1e-3 relative, parameter gradients within 1e-3 (norm)."""
    from dvs_of_training_framework_amd.loss import init_losses
    from dvs_of_training_framework_amd.options import Mish
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch
    B, H, W, C = 2, 32, 48, 5
    model = make_model(C, activation=Mish() if mish else torch.nn.ReLU())
    model.train()
    ev = init_losses((H, W), B, model, DEV, sequence_length=1)
    b_np = synthetic.make_batch(11, B, H, W, 1500)
    batch = synthetic.to_torch(b_np, DEV)
    loss, terms, tags, extra = process_minibatch(
        model, batch, FakeTimer(), DEV, True, ev, [0.5, 1, 1], return_prediction=True)
    loss.backward()
    # CPU port
    state = {k[len('predictor.'):]: v.detach().cpu().clone().requires_grad_(True)
             for k, v in model.state_dict().items()}
    grid, _, _ = orc.voxelize(b_np['events'], np.zeros(B, np.float32),
                              np.full(B, 0.04, np.float32), B, C, H, W)
    flows = ref_predictor(state, torch.from_numpy(grid), mish)
    o_terms, o_loss, o_grads = orc.losses(
        [f.detach().numpy() for f in flows], b_np['timestamps'].reshape(B, 2), np.arange(B),
        b_np['images'], b_np['timestamps'], b_np['sample_idx'])
    torch.autograd.backward(flows, [torch.from_numpy(g) for g in o_grads])
    assert abs(float(loss.detach()) - o_loss) <= 1e-3 * abs(o_loss)
    got_terms = np.array([list(t) for t in terms])
    np.testing.assert_allclose(got_terms, o_terms, rtol=1e-3, atol=1e-6)
    for f, r in zip(extra['prediction'], flows):
        assert (f.detach().cpu() - r.detach()).abs().max() <= 1e-3 * r.abs().max() + 1e-6
    for name, p in model.named_parameters():
        r = state[name[len('predictor.'):]].grad
        g = p.grad.detach().cpu()
        assert (g - r).norm() <= 2e-3 * r.norm() + 1e-9, name
    assert list(tags) == ['4x6', '8x12', '16x24', '32x48']


def test_unit_backward_equals_loss_backward():
    """loss.unit_backward (cached device 1.0 seed recognised by the fused loss:
    no ones_like fill, no x1.0 pass) gives the gradients of loss.backward()
    (x1.0 is exact; the comparison allows for the voxeliser's float atomics,
    whose summation order differs from run to run); a scaled loss still goes
    through the scaling pass."""
    from dvs_of_training_framework_amd.loss import init_losses, unit_backward
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch
    B, H, W, C = 2, 32, 48, 5
    model = make_model(C)
    model.train()
    ev = init_losses((H, W), B, model, DEV, sequence_length=1)
    batch = synthetic.to_torch(synthetic.make_batch(5, B, H, W, 1500), DEV)

    def grads(how):
        for p in model.parameters():
            p.grad = None
        loss, _, _ = process_minibatch(model, batch, FakeTimer(), DEV, True, ev, [0.5, 1, 1])
        how(loss)
        return [p.grad.detach().clone() for p in model.parameters()]
    ref = grads(lambda l: l.backward())
    got = grads(unit_backward)
    half = grads(lambda l: (l * 0.5).backward())
    for a, b, c in zip(ref, got, half):
        assert (a - b).norm() <= 1e-4 * a.norm() + 1e-12
        assert (c - 0.5 * a).norm() <= 1e-4 * a.norm() + 1e-12


def test_fused_adamw_matches_torch():
    from dvs_of_training_framework_amd.optim import FusedAdamW
    torch.manual_seed(3)
    shapes = [(64, 5, 3, 3), (64,), (7,), (130, 33, 3, 3), (4097,)]
    ps = [torch.randn(s) for s in shapes]
    a = [p.clone().cuda().requires_grad_(True) for p in ps]
    a[0] = a[0].detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = [p.clone().requires_grad_(True) for p in ps]
    fo = FusedAdamW([{'params': a[:2], 'lr': 1e-2}, {'params': a[2:]}], lr=1e-3,
                    weight_decay=1e-2, amsgrad=True)
    to = torch.optim.AdamW([{'params': b[:2], 'lr': 1e-2}, {'params': b[2:]}], lr=1e-3,
                           weight_decay=1e-2, amsgrad=True)
    for step in range(4):
        for x, y in zip(a, b):
            g = torch.randn(y.shape) * (1 + step)
            y.grad = g.clone()
            x.grad = g.cuda().contiguous(memory_format=torch.channels_last) \
                if x.dim() == 4 else g.cuda()
            if x.dim() == 4 and not x.is_contiguous(memory_format=torch.channels_last):
                x.grad = g.cuda()
        fo.step()
        to.step()
        for x, y in zip(a, b):
            assert (x.detach().cpu() - y.detach()).abs().max() <= 2e-6 * y.abs().max()
    sd = fo.state_dict()
    assert set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'}


def test_fused_adamw_resumes_from_a_torch_adamw_state_dict():
    """Checkpoint interchange (utils/serializer.py:60-110 stores
    optimizer.state_dict()): state written by torch.optim.AdamW(amsgrad=True)
    holds CONTIGUOUS moments; the conv weights here are channels_last.
    Loading re-lays the state to the parameter's strides and the next steps
    match torch's."""
    from dvs_of_training_framework_amd.optim import FusedAdamW
    torch.manual_seed(8)
    shapes = [(64, 5, 3, 3), (64,), (2, 32, 1, 1)]
    ps = [torch.randn(s) for s in shapes]
    b = [p.clone().requires_grad_(True) for p in ps]
    to = torch.optim.AdamW(b, lr=1e-3, weight_decay=1e-2, amsgrad=True)
    gs = [[torch.randn(s) for s in shapes] for _ in range(4)]
    for step in range(2):
        for y, g in zip(b, gs[step]):
            y.grad = g.clone()
        to.step()
    sd = to.state_dict()
    a = []
    for y in b:
        q = y.detach().clone().cuda()
        if q.dim() == 4:
            q = q.contiguous(memory_format=torch.channels_last)
        a.append(q.requires_grad_(True))
    fo = FusedAdamW(a, lr=1e-3, weight_decay=1e-2, amsgrad=True)
    fo.load_state_dict(sd)
    st = fo.state[a[0]]
    assert st['exp_avg'].stride() == a[0].stride() and st['step'] == 2
    for step in range(2, 4):
        for x, y, g in zip(a, b, gs[step]):
            y.grad = g.clone()
            gx = g.cuda()
            x.grad = gx.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else gx
        fo.step()
        to.step()
        for x, y in zip(a, b):
            assert (x.detach().cpu() - y.detach()).abs().max() <= 2e-6 * y.abs().max()
    # and back: torch loads what the fused optimizer saved
    to2 = torch.optim.AdamW([y.detach().clone().requires_grad_(True) for y in b], lr=1e-3,
                            weight_decay=1e-2, amsgrad=True)
    to2.load_state_dict(fo.state_dict())


def test_train_loop_reduces_loss_and_accumulates():
    import train_flownet as tf
    from dvs_of_training_framework_amd.loss import init_losses
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import train
    args = tf.parse_args(['-m', '/tmp/dvsof_test_model', '--optimizer', 'ADAM', '-bs', '4',
                          '-mbs', '2', '--height', '32', '--width', '32', '-lr', '1e-3',
                          '--event-representation-depth', '3', '--synthetic',
                          '--synthetic-events', '500', '-ne', '6', '-d', 'cuda:0'])
    model = make_model(3)
    optimizer, scheduler = tf.construct_train_tools(args, model)
    assert type(optimizer).__name__ == 'FusedAdamW' and len(optimizer.param_groups) == 1
    ev = init_losses(args.shape, args.mbs, model, DEV, sequence_length=1)

    class Log:
        rows = []

        def add_scalar(self, t, v, x):
            self.rows.append((t, v, x))
    fixed = synthetic.make_batch(5, 2, 32, 32, 500)
    loader = (synthetic.to_torch(fixed) for _ in range(100))
    w0 = model.predictor.dec[3].flow.weight.detach().clone()
    train(model, DEV, loader, optimizer, args.training_steps, scheduler, Log(), ev,
          accumulation_steps=args.accum_step, timers=FakeTimer())
    losses = [v for t, v, x in Log.rows if t == 'General/Train loss']
    assert len(losses) == 6 and all(np.isfinite(losses))
    assert losses[-1] < losses[0]
    assert not torch.equal(w0, model.predictor.dec[3].flow.weight)
    assert [x for t, v, x in Log.rows if t == 'General/Train loss'] == [4, 8, 12, 16, 20, 24]


def test_cli_with_capture_accumulation_and_device_feeder(tmp_path):
    """train_flownet.main end to end on synthetic batches: --capture with
    bs // mbs = 2 micro-batches per optimizer step (the roles of capture.py),
    batches moved by feed.DeviceFeeder; the checkpoint the reference's
    serializer would load is written (utils/serializer.py:60-110 schema)."""
    import train_flownet as tf
    out = tmp_path / 'model'
    pkg = Path(tf.__file__).resolve().parent / 'dvs_of_training_framework_amd'
    tf.main(['-m', str(out), '--flownet_path', str(pkg), '--optimizer', 'ADAM', '-bs', '4', '-mbs', '2', '--height', '64',
             '--width', '64', '-lr', '1e-3', '--event-representation-depth', '3', '--synthetic',
             '--synthetic-events', '3000', '-ne', '5', '-d', 'cuda:0', '--capture',
             '--device-feeder'])
    ckpt = torch.load(out / 'step_5.pt', weights_only=True)
    assert set(ckpt) >= {'model', 'optimizer', 'global_step'} and ckpt['global_step'] == 5
    steps = {int(v['step']) for v in ckpt['optimizer']['state'].values()}
    assert steps == {5}         # one AdamW update per optimizer step, replayed or not
    assert all(torch.isfinite(v).all() for v in ckpt['model'].values())


def test_optical_flow_wrapper(tmp_path):
    # DummyNet/of.py:53-74,120-125 contract: NHWC numpy out
    from dvs_of_training_framework_amd import OpticalFlow
    model = make_model(9)
    path = tmp_path / 'model.pth'
    torch.save({'model': model.state_dict()}, path)
    of = OpticalFlow((64, 80), model=str(path), device=torch.device(DEV))
    rng = np.random.default_rng(0)
    evs = [(rng.integers(0, 80, 300), rng.integers(0, 64, 300),
            np.sort(rng.random(300)) * 0.05 + 10.0, rng.integers(0, 2, 300) * 2 - 1)
           for _ in range(2)]
    out = of(evs, [10.0, 10.0], [10.05, 10.05])
    assert isinstance(out, np.ndarray) and out.shape == (2, 64, 80, 2)
    allf = of(evs, [10.0, 10.0], [10.05, 10.05], return_all=True)
    assert [a.shape for a in allf] == [(2, 8, 10, 2), (2, 16, 20, 2), (2, 32, 40, 2),
                                       (2, 64, 80, 2)]


@pytest.mark.parametrize('kind', ['radam', 'ranger'])
def test_fused_radam_ranger_match_restatement(kind):
    """Through the un-rectified early phase (N_sma small), the rectified phase
    and (Ranger) two Lookahead syncs; weights channels_last like the model's."""
    from dvs_of_training_framework_amd.optim import FusedRAdam, FusedRanger
    from oracle.ref_optim import RefRAdam, RefRanger
    torch.manual_seed(5)
    shapes = [(32, 5, 3, 3), (32,), (64, 130, 3, 3), (2, 32, 1, 1), (4099,)]
    ps = [torch.randn(s) * 0.1 for s in shapes]
    a = []
    for p in ps:
        q = p.clone().cuda()
        if q.dim() == 4:
            q = q.contiguous(memory_format=torch.channels_last)
        a.append(q.requires_grad_(True))
    b = [p.clone().requires_grad_(True) for p in ps]
    if kind == 'radam':
        fo, ro = FusedRAdam(a, lr=2e-3, weight_decay=1e-2), RefRAdam(b, lr=2e-3, weight_decay=1e-2)
    else:
        fo, ro = FusedRanger(a, lr=2e-3, weight_decay=1e-2), RefRanger(b, lr=2e-3, weight_decay=1e-2)
    for step in range(13):
        for x, y in zip(a, b):
            g = torch.randn(y.shape) * (1 + 0.1 * step)
            y.grad = g.clone()
            gx = g.cuda()
            x.grad = gx.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else gx
        fo.step()
        ro.step()
        for x, y in zip(a, b):
            err = (x.detach().cpu() - y.detach()).abs().max()
            assert err <= 5e-6 * y.abs().max() + 1e-7, (kind, step, float(err))
    assert set(fo.state_dict()['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq', 'slow_buffer'}


def test_predictor_full_size_vs_float64_and_determinism():
    """BASELINE config-2 size (B=8, 256x256x5).  Reference = the same
    restatement evaluated in float64 on the host.  Flows within 1e-3 of the
    peak.  Parameter gradients: any fp32 implementation differs from float64 by
    ReLU-mask flips of activations within ~1e-6 of zero, so the pin is on the
    field norm (3e-3); measured on MI355X the HIP path is at 0.3e-3..1.5e-3,
    ATen/MIOpen's own GPU fp32 path at 0.3e-3..4.7e-3 (tools/dbg_fullsize.py).
    Two runs of the HIP path are bitwise identical (fixed-order slab
    reductions, no float atomics in the conv stack)."""
    from dvs_of_training_framework_amd.predictor import Predictor
    torch.manual_seed(2)
    B, Cin, H, W = 8, 5, 256, 256
    net = Predictor(Cin).cuda()
    x = torch.randn(B, Cin, H, W, device=DEV)
    gfl = [torch.randn(B, 2, H // s, W // s, device=DEV) * (0.5 / s) for s in (8, 4, 2, 1)]

    def run_hip():
        for p in net.parameters():
            p.grad = None
        flows = net(x)
        torch.autograd.backward(flows, gfl)
        return [f.detach().clone() for f in flows], \
            {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    f1, g1 = run_hip()
    f2, g2 = run_hip()
    for a, b in zip(f1, f2):
        assert torch.equal(a, b)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    state = {k: v.detach().cpu().double().contiguous().requires_grad_(True)
             for k, v in net.state_dict().items()}
    ref = ref_predictor(state, x.cpu().double())
    torch.autograd.backward(ref, [g.cpu().double() for g in gfl])
    for a, r in zip(f1, ref):
        r = r.detach().float().cuda()
        assert (a - r).abs().max() <= 1e-3 * r.abs().max()
    for n, g in g1.items():
        r = state[n].grad.float().cuda()
        assert (g - r).norm() <= 3e-3 * r.norm() + 1e-9, n
        assert (g - r).abs().max() <= 2e-2 * r.abs().max() + 1e-9, n


def test_bf16_operand_mode_tracks_the_f32_predictor():
    """compute_dtype='bf16' (operands rounded in registers, f32 accumulate and
    storage) against the exact-f32 predictor with the same weights: flows and
    parameter gradients agree to bf16 accuracy."""
    from dvs_of_training_framework_amd.predictor import Predictor
    torch.manual_seed(11)
    a = Predictor(5).cuda()
    b = Predictor(5, compute_dtype='bf16').cuda()
    b.load_state_dict(a.state_dict())
    x = torch.randn(2, 5, 64, 64, device='cuda')
    fa, fb = a(x), b(x)
    seeds = [torch.randn_like(f) for f in fa]
    torch.autograd.backward(fa, seeds)
    torch.autograd.backward(fb, seeds)
    for u, v in zip(fa, fb):
        assert float((u - v).norm()) <= 3e-2 * float(u.norm())
    # gradients cross 16 bf16-rounded layers and ReLU boundaries: direction
    # preserved (cosine), magnitude error of a few per cent per tensor
    for p, q in zip(a.parameters(), b.parameters()):
        assert bool(torch.isfinite(q.grad).all())
        rel = float((p.grad - q.grad).norm()) / (float(p.grad.norm()) + 1e-12)
        cos = float((p.grad * q.grad).sum()) / (float(p.grad.norm() * q.grad.norm()) + 1e-12)
        assert rel <= 0.2 and cos >= 0.98, (rel, cos)


def test_bf16x3_mode_is_f32_accurate():
    """compute_dtype='bf16x3' (hi/lo split operands, three bf16 products) stays
    inside the north star's 1e-3 relative parity budget with a wide margin."""
    from dvs_of_training_framework_amd.predictor import Predictor
    torch.manual_seed(12)
    a = Predictor(5).cuda()
    b = Predictor(5, compute_dtype='bf16x3').cuda()
    b.load_state_dict(a.state_dict())
    x = torch.randn(2, 5, 64, 64, device='cuda')
    fa, fb = a(x), b(x)
    seeds = [torch.randn_like(f) for f in fa]
    torch.autograd.backward(fa, seeds)
    torch.autograd.backward(fb, seeds)
    for u, v in zip(fa, fb):
        assert float((u - v).norm()) <= 1e-4 * float(u.norm())
    for p, q in zip(a.parameters(), b.parameters()):
        assert float((p.grad - q.grad).norm()) <= 1e-3 * float(p.grad.norm()) + 1e-9


def test_bf16_twins_mode_matches_the_bf16_operand_mode():
    """compute_dtype='bf16s' (bf16 twins of activations, gradients and
    prepared weights streamed through LDS) against 'bf16' (f32 tensors,
    operands rounded in registers).  The forward of a layer is the SAME
    arithmetic in both -- bf16(x) * bf16(w) accumulated in f32 -- so flows agree
    to accumulation order; the data gradients differ by one bf16 rounding of
    the stored gradient per layer.  Mish exercises the z twin-less path too."""
    from dvs_of_training_framework_amd.options import Mish
    from dvs_of_training_framework_amd.predictor import Predictor
    for act in (torch.nn.ReLU(), Mish()):
        torch.manual_seed(13)
        a = Predictor(5, activation=act, compute_dtype='bf16').cuda()
        b = Predictor(5, activation=act, compute_dtype='bf16s').cuda()
        b.load_state_dict(a.state_dict())
        x = torch.randn(2, 5, 64, 96, device='cuda')
        fa, fb = a(x), b(x)
        seeds = [torch.randn_like(f) for f in fa]
        torch.autograd.backward(fa, seeds)
        torch.autograd.backward(fb, seeds)
        for u, v in zip(fa, fb):
            assert float((u - v).norm()) <= 2e-3 * float(u.norm())
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            assert bool(torch.isfinite(q.grad).all()), n
            rel = float((p.grad - q.grad).norm()) / (float(p.grad.norm()) + 1e-12)
            assert rel <= 5e-2, (n, rel)


@pytest.mark.parametrize('dtype', ['f32', 'bf16x3', 'bf16', 'bf16s'])
def test_training_reduces_the_loss_on_a_fixed_batch(dtype):
    """End-to-end sanity of the whole step (voxelise, predictor, fused loss,
    two-stream backward, fused AdamW): 40 steps on one batch lower the loss."""
    from dvs_of_training_framework_amd import synthetic
    from dvs_of_training_framework_amd.loss import init_losses
    from dvs_of_training_framework_amd.net import Model
    from dvs_of_training_framework_amd.optim import FusedAdamW
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch
    torch.manual_seed(5)
    B, H, W = 2, 64, 64
    model = Model('cuda', event_representation_depth=5, compute_dtype=dtype)
    model.train()
    opt = FusedAdamW(model.predictor.parameters(), lr=1e-3, weight_decay=1e-4, amsgrad=True)
    ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
    batch = synthetic.to_torch(synthetic.make_batch(7, B, H, W, 4096), 'cuda')
    losses = []
    for _ in range(40):
        loss, _, _ = process_minibatch(model, batch, FakeTimer(), 'cuda', True, ev, [0.5, 1, 1])
        loss.backward()
        model.strict = False
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert np.mean(losses[-5:]) < 0.97 * np.mean(losses[:5]), (losses[:5], losses[-5:])


@pytest.mark.parametrize('flush_at', [None, (5,)])
@pytest.mark.parametrize('opt_name', ['adamw', 'ranger'])
def test_optimizer_fused_into_backward_is_the_same_update(opt_name, flush_at):
    """optim.fuse_into_backward: buckets are updated during the backward (one
    launch per bucket, or everything up to a flush bucket in one launch); the
    weights after step() are bit-identical to the plain step()."""
    from dvs_of_training_framework_amd.optim import FusedAdamW, FusedRanger
    from dvs_of_training_framework_amd.predictor import Predictor

    def run(fused):
        torch.manual_seed(21)
        net = Predictor(5).cuda()
        opt = FusedAdamW(net.parameters(), lr=1e-3, weight_decay=1e-4, amsgrad=True) \
            if opt_name == 'adamw' else FusedRanger(net.parameters(), lr=1e-3)
        if fused:
            opt.fuse_into_backward(net, flush_at=flush_at)
        g = torch.Generator(device='cuda').manual_seed(3)
        for it in range(3):
            x = torch.randn(2, 5, 64, 64, device='cuda', generator=g)
            flows = net(x)
            seeds = [torch.randn(f.shape, device='cuda', generator=g) for f in flows]
            torch.autograd.backward(flows, seeds)
            opt.step()
            opt.zero_grad(set_to_none=True)
        return [p.detach().clone() for p in net.parameters()]
    a, b = run(False), run(True)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize('B,C,H,W,n', [
    (2, 9, 480, 640, 100000),      # BASELINE.json configs[3] shape (MVSEC-like)
    (1, 12, 512, 512, 300000),     # configs[4] shape
    (2, 5, 260, 346, 20000),       # DAVIS frame size of the reference fixtures (padded to 272x352)
])
def test_one_training_step_at_other_baseline_shapes(B, C, H, W, n):
    """Whole step at the other configured shapes: finite loss and gradients,
    flows of the requested size, second run of the same step bit-identical
    in the predictor (the voxeliser's float atomics aside: same grid reused)."""
    from dvs_of_training_framework_amd import synthetic
    from dvs_of_training_framework_amd.loss import init_losses
    from dvs_of_training_framework_amd.net import Model
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch
    torch.manual_seed(9)
    model = Model('cuda', event_representation_depth=C)
    model.train()
    ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
    batch = synthetic.to_torch(synthetic.make_batch(17, B, H, W, n), 'cuda')
    loss, terms, tags, info = process_minibatch(model, batch, FakeTimer(), 'cuda', True, ev,
                                                [0.5, 1, 1], return_prediction=True)
    assert [tuple(f.shape) for f in info['prediction']] == \
        [(B, 2, H // s, W // s) for s in (8, 4, 2, 1)]
    assert list(tags) == [f'{H // s}x{W // s}' for s in (8, 4, 2, 1)]
    loss.backward()
    assert np.isfinite(float(loss.detach()))
    for p in model.predictor.parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all())


def test_optical_flow_graph_replay_matches_eager_calls():
    """OpticalFlow(graph=True): the captured HIP graph (voxelise + weight
    forms + predictor) reproduces the eager call, also when a later call has
    fewer events than the captured capacity (padding with x = y = -1)."""
    from dvs_of_training_framework_amd.of import OpticalFlow
    H = W = 64
    rng = np.random.default_rng(3)

    def events(n):
        return [(rng.integers(0, W, n), rng.integers(0, H, n),
                 np.sort(rng.random(n) * 0.04), rng.integers(0, 2, n) * 2 - 1)
                for _ in range(2)]
    torch.manual_seed(4)
    eager = OpticalFlow((H, W), model=None, event_representation_depth=5)
    graph = OpticalFlow((H, W), model=None, graph=True, event_representation_depth=5)
    graph.load_state_dict(eager._net.state_dict())
    for n in (6000, 6000, 2500, 5999):
        ev = events(n)
        a = eager(ev, [0.0, 0.0], [0.04, 0.04])
        b = graph(ev, [0.0, 0.0], [0.04, 0.04])
        assert a.shape == b.shape == (2, H, W, 2)
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(a).max())
    assert len(graph._graphs) == 1 and graph._graphs[2]['cap'] == 16384
