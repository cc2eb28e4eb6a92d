"""Parity of the HIP loss path (through the C ABI) with the CPU oracle and
with the reference goldens.  Tolerances: loss terms 1e-4 relative (north star
budget: 1e-3), gradient fields 1e-3 in L2 norm / 2e-3 of the peak in max-abs
(see tests/test_oracle_loss.py for why max-abs is looser)."""
import numpy as np
import pytest
import torch

from oracle import cpu_oracle as orc
from tests.cases import SYNTH_CASES, synth_case, fixture_case

pytestmark = pytest.mark.gpu

TERM_RTOL = 1e-4
WEIGHTS = (0.5, 1.0, 1.0)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_hip(c, fused):
    from dvs_of_training_framework_amd.loss import Losses
    ev = Losses(c['shapes'], c['B'], 'cuda')
    flows = [dev(f).requires_grad_(True) for f in c['flows']]
    args = (flows, dev(c['flow_ts']), dev(c['flow_sample_idx']),
            dev(c['images']), dev(c['timestamps']), dev(c['sample_idx']))
    if fused:
        loss, terms = ev.fused(*args, weights=WEIGHTS)
        terms = terms.cpu().numpy().astype(np.float64)
    else:
        t = ev(*args)
        assert len(t) == 3 and all(len(x) == len(flows) for x in t)
        loss = sum(w * sum(x) / len(x) for x, w in zip(t, WEIGHTS))
        terms = np.array([[float(v) for v in x] for x in t])
    loss.backward()
    return terms, float(loss), [f.grad.cpu().numpy() for f in flows]


def check_grads(got, want):
    for k, (g, r) in enumerate(zip(got, want)):
        assert np.linalg.norm(g - r) <= 1e-3 * np.linalg.norm(r) + 1e-12, k
        assert np.abs(g - r).max() <= 2e-3 * np.abs(r).max() + 1e-12, k


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('name', list(SYNTH_CASES))
def test_synthetic_vs_oracle_and_reference(golden_loss, name, fused):
    c = synth_case(name)
    terms, loss, grads = run_hip(c, fused)
    o_terms, o_loss, o_grads = orc.losses(
        c['flows'], c['flow_ts'], c['flow_sample_idx'], c['images'],
        c['timestamps'], c['sample_idx'], weights=WEIGHTS)
    np.testing.assert_allclose(terms, o_terms, rtol=TERM_RTOL, atol=1e-7)
    np.testing.assert_allclose(terms, golden_loss[f'{name}_terms'],
                               rtol=TERM_RTOL, atol=1e-7)
    assert abs(loss - golden_loss[f'{name}_loss']) <= TERM_RTOL * abs(loss)
    check_grads(grads, o_grads)
    check_grads(grads, [golden_loss[f'{name}_grad{k}']
                        for k in range(len(grads))])


def test_reference_golden_triples(fixtures, golden_loss):
    # /root/reference/tests/loss/test_loss.py:25-65, tolerance 5e-6 as there
    t, _, _ = run_hip(fixture_case(fixtures, 1, False), False)
    for v, gt in zip(t[:, 0], [0.002, 0.622660, 0]):
        assert abs(v - gt) < 5e-6
    t, loss, g = run_hip(fixture_case(fixtures, 1, True), False)
    for v, gt in zip(t[:, 0], [0.002120, 0.652659, 0.007802]):
        assert abs(v - gt) < 5e-6
    check_grads(g, [golden_loss['fixture1_pred_grad']])


def test_no_changes():
    # /root/reference/tests/loss/test_loss.py:8-22
    from dvs_of_training_framework_amd.loss import Losses
    z = torch.zeros
    ev = Losses([(5, 6)], 1, 'cuda')
    ts = torch.tensor([0, 0.4]).cuda()
    t = ev([z(1, 2, 5, 6).cuda()], ts.view(1, 2), z(1, dtype=torch.long).cuda(),
           z(2, 1, 5, 6).cuda(), ts, z(2, dtype=torch.long).cuda())
    for l, gt in zip(t, [0.002, 0.002, 0]):
        assert len(l) == 1 and abs(float(l[0]) - gt) < 5e-6


@pytest.mark.parametrize('i', range(10))
def test_fixture_table(fixtures, golden_loss, i):
    t0, _, _ = run_hip(fixture_case(fixtures, i, False), True)
    t1, _, _ = run_hip(fixture_case(fixtures, i, True), False)
    np.testing.assert_allclose(t0[:, 0], golden_loss['fixture_zero_terms'][i],
                               rtol=TERM_RTOL, atol=1e-8)
    np.testing.assert_allclose(t1[:, 0], golden_loss['fixture_pred_terms'][i],
                               rtol=TERM_RTOL, atol=1e-8)


def test_single_scale_loss_object():
    # Loss.__call__ surface, utils/loss.py:121-171
    from dvs_of_training_framework_amd.loss import Loss
    rng = np.random.default_rng(5)
    prev = rng.random((3, 1, 20, 30), dtype=np.float32) * 255
    nxt = rng.random((3, 1, 20, 30), dtype=np.float32) * 255
    flow = (rng.standard_normal((3, 2, 20, 30)) * 4).astype(np.float32)
    s, p, b = Loss((20, 30), 4, 'cuda')(dev(prev), dev(nxt), dev(flow))
    want, _ = orc.loss_scale_fwd(prev, nxt, flow)
    np.testing.assert_allclose([float(s), float(p), float(b)], want,
                               rtol=TERM_RTOL)


def test_frame_resolution_asserts_like_reference():
    from dvs_of_training_framework_amd.loss import Losses
    c = synth_case('cfg1_64')
    ev = Losses(c['shapes'], c['B'], 'cuda')
    bad_ts = c['flow_ts'].copy()
    bad_ts[0, 0] = 0.0123
    with pytest.raises(AssertionError):
        ev([dev(f) for f in c['flows']], dev(bad_ts),
           dev(c['flow_sample_idx']), dev(c['images']), dev(c['timestamps']),
           dev(c['sample_idx']))


def test_baseline_size_properties():
    """Config 2 size (B=8, 256x256, 4 scales): size-independent properties.
    fused == separate; backward is linear in the seeds; zero flow gives the
    closed forms smooth = (1e-6)^0.45 and border = 0."""
    from dvs_of_training_framework_amd import synthetic
    from dvs_of_training_framework_amd.loss import Losses
    B, H, W = 8, 256, 256
    shapes = synthetic.scale_shapes(H, W)
    batch = synthetic.make_batch(99, B, H, W, events_per_sample=0)
    flows_np = synthetic.make_flows(100, B, shapes, 2.0)
    c = dict(shapes=shapes, B=B, flows=flows_np,
             flow_ts=batch['timestamps'].reshape(B, 2),
             flow_sample_idx=np.arange(B), images=batch['images'],
             timestamps=batch['timestamps'], sample_idx=batch['sample_idx'])
    ta, la, ga = run_hip(c, False)
    tb, lb, gb = run_hip(c, True)
    np.testing.assert_allclose(ta, tb, rtol=1e-6)
    assert abs(la - lb) <= 1e-5 * abs(la)
    for a, b in zip(ga, gb):
        assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max() + 1e-12
    # linearity in the seeds
    ev = Losses(shapes, B, 'cuda')
    args = (dev(c['flow_ts']), dev(c['flow_sample_idx']), dev(c['images']),
            dev(c['timestamps']), dev(c['sample_idx']))

    def grads_for(seed):
        flows = [dev(f).requires_grad_(True) for f in flows_np]
        t = ev(flows, *args)
        tot = sum(s * sum(x) for x, s in zip(t, seed))
        tot.backward()
        return [f.grad for f in flows]
    g1, g2, g3 = grads_for((1, 0, 0)), grads_for((0, 1, 0)), grads_for((0, 0, 1))
    g = grads_for((0.3, 2.0, -1.5))
    for a, x, y, z in zip(g, g1, g2, g3):
        want = 0.3 * x + 2.0 * y - 1.5 * z
        assert float((a - want).abs().max()) <= 1e-5 * float(want.abs().max())
    # zero flow
    zf = [torch.zeros(B, 2, h, w, device='cuda') for h, w in shapes]
    t = ev(zf, *args)
    for k in range(4):
        assert abs(float(t[0][k]) - 1e-6 ** 0.45) < 1e-8
        assert float(t[2][k]) == 0.0
    # the full-size result against the oracle: all four scales, terms AND flow gradients
    o_terms, o_loss, o_grads = orc.losses(flows_np, c['flow_ts'], c['flow_sample_idx'],
                                          c['images'], c['timestamps'], c['sample_idx'])
    np.testing.assert_allclose(ta, o_terms, rtol=TERM_RTOL)
    assert abs(la - o_loss) <= TERM_RTOL * abs(o_loss)
    for a, o in zip(ga, o_grads):       # field norm
        assert np.linalg.norm(a - o) <= 1e-3 * np.linalg.norm(o), \
            np.linalg.norm(a - o) / np.linalg.norm(o)


@pytest.mark.parametrize('src,shapes,D', [
    ((256, 256), [(32, 32), (64, 64), (128, 128), (256, 256)], 6),     # config 2
    ((260, 346), [(33, 44), (65, 87), (130, 173), (260, 346)], 6),     # odd sizes
    ((480, 640), [(60, 80), (120, 160), (240, 320), (480, 640)], 6),   # config 4
    ((64, 64), [(64, 64)], 6),                                         # one level
    ((96, 80), [(12, 10), (12, 10), (48, 40)], 6),                     # equal levels
    ((64, 64), [(32, 32), (16, 16)], 6),                               # shrinking: per-level path
    # many frames: 32 x 128 tiles of the finest level, 16-byte stores (batch 16 = 32 frames
    # at the benchmark size; ragged right / bottom tiles at 480 x 640; a width that is no
    # multiple of 4 takes the scalar stores)
    ((256, 256), [(32, 32), (64, 64), (128, 128), (256, 256)], 32),
    ((480, 640), [(60, 80), (120, 160), (240, 320), (480, 640)], 8),
    ((260, 346), [(33, 44), (65, 87), (130, 173), (260, 346)], 24),
])
def test_pyramid_one_launch_is_bitwise_the_cascade(src, shapes, D):
    """dvsof_loss_pyramid == K dependent dvsof_resize_bilinear_ac calls
    (utils/loss.py:207-210), bit for bit, and == torch's interpolate within
    float rounding."""
    from dvs_of_training_framework_amd.loss import Losses, interpolate
    g = torch.Generator().manual_seed(5)
    img = (torch.rand(D, 1, *src, generator=g) * 255).cuda()
    ev = Losses(shapes, D // 2, 'cuda')
    flows = [torch.zeros(D // 2, 2, h, w, device='cuda') for h, w in shapes]
    levels = ev._pyramid(flows, img)
    cur, ref = img[:, 0], img
    for lev, shape in zip(levels, shapes):
        cur = interpolate(cur, shape)
        assert torch.equal(lev, cur), shape
        ref = torch.nn.functional.interpolate(ref, size=shape, mode='bilinear',
                                              align_corners=True)
        assert float((lev - ref[:, 0]).abs().max()) <= 1e-3
