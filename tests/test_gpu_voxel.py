"""Event binning on the GPU: integer parts bit-exact against the oracle
(count image; temporal bin and linear voxel index), float sums within 1e-5
absolute per accumulated event (float atomics are order dependent)."""
import numpy as np
import pytest
import torch

from oracle import cpu_oracle as orc
from dvs_of_training_framework_amd import synthetic

pytestmark = pytest.mark.gpu


def dev_events(ev):
    return {k: torch.from_numpy(v).cuda() for k, v in ev.items()}


@pytest.mark.parametrize('B,C,H,W,n', [(2, 3, 64, 64, 4096), (8, 5, 256, 256, 65536),
                                       (1, 9, 37, 53, 1000), (3, 12, 16, 16, 0)])
def test_voxelize_vs_oracle(B, C, H, W, n):
    from dvs_of_training_framework_amd.voxel import voxelize
    rng = np.random.default_rng(7)
    ev = synthetic.make_events(rng, B, H, W, n)
    t0 = np.zeros(B, np.float32)
    t1 = np.full(B, synthetic.WINDOW, np.float32)
    if n:  # a few events outside the window / frame must be dropped
        ev['timestamp'][::97] += 1.0
        ev['x'][::89] = W + 3
    want, bin0, lin0 = orc.voxelize(ev, t0, t1, B, C, H, W)
    got, gbin, glin = voxelize(dev_events(ev), torch.from_numpy(t0).cuda(),
                               torch.from_numpy(t1).cuda(), B, C, H, W,
                               debug=True)
    assert got.shape == (B, C, H, W) and got.dtype == torch.float32
    assert np.array_equal(gbin.cpu().numpy(), bin0)      # bit-exact
    assert np.array_equal(glin.cpu().numpy(), lin0)      # bit-exact
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize('B,n', [(1, 1_000_000),
                                 (3, 800_000)])    # 2.4 M events: the 8-events-per-thread pass
def test_voxel_mass_conservation_full_size(B, n):
    """Config-5 scale input (1M events per sample, 512x512x12): sum over bins of
    the grid equals the polarity-signed count image (weights (1-f) + f = 1)."""
    from dvs_of_training_framework_amd.voxel import voxelize
    C, H, W = 12, 512, 512
    rng = np.random.default_rng(8)
    ev = synthetic.make_events(rng, B, H, W, n)
    d = dev_events(ev)
    t0 = torch.zeros(B, device='cuda')
    t1 = torch.full((B,), synthetic.WINDOW, device='cuda')
    grid = voxelize(d, t0, t1, B, C, H, W)
    signed = torch.zeros(B * H * W, device='cuda')
    signed.index_add_(0, (d['sample_index'] * H + d['y']) * W + d['x'], d['polarity'].float())
    err = (grid.sum(1).view(-1) - signed).abs().max()
    assert float(err) < 1e-3


def test_count_image_bit_exact():
    from dvs_of_training_framework_amd.voxel import get_count_image
    rng = np.random.default_rng(9)
    for H, W, n in [(260, 346, 20000), (8, 8, 5), (64, 64, 0)]:
        x = rng.integers(0, W, n)
        y = rng.integers(0, H, n)
        got = get_count_image([x, y], (H, W))
        assert got.dtype == np.uint64
        assert np.array_equal(got, orc.count_image(x, y, H, W))
    with pytest.raises(ValueError):
        get_count_image([np.array([W]), np.array([0])], (H, W))


def test_count_image_fixture(fixtures):
    # events of the reference's own test sequence (tests/data/seq/000001.hdf5)
    from dvs_of_training_framework_amd.voxel import get_count_image
    ev = fixtures['events_1']
    got = get_count_image([ev[:, 0], ev[:, 1]], (260, 346))
    assert got.sum() == ev.shape[0]
    assert np.array_equal(got, orc.count_image(ev[:, 0].astype(np.int64),
                                               ev[:, 1].astype(np.int64),
                                               260, 346))


def test_voxelize_tiled_overflow_and_ragged():
    """LDS-tiled path: all events of one sample crowd into a few pixels (bucket
    overflow -> overflow list), another sample is empty, frame not a multiple
    of the tile; integer parts bit-exact, sums within the float-atomics budget."""
    from dvs_of_training_framework_amd.voxel import voxelize
    B, C, H, W, n = 3, 5, 70, 90, 30000
    rng = np.random.default_rng(21)
    ev = synthetic.make_events(rng, B, H, W, n)
    crowd = ev['sample_index'] == 0
    ev['x'][crowd] = rng.integers(3, 6, crowd.sum())
    ev['y'][crowd] = rng.integers(40, 42, crowd.sum())
    keep = ev['sample_index'] != 1                 # sample 1 gets no events
    ev = {k: v[keep] for k, v in ev.items()}
    t0 = np.zeros(B, np.float32)
    t1 = np.full(B, synthetic.WINDOW, np.float32)
    want, bin0, lin0 = orc.voxelize(ev, t0, t1, B, C, H, W)
    got, gbin, glin = voxelize(dev_events(ev), torch.from_numpy(t0).cuda(),
                               torch.from_numpy(t1).cuda(), B, C, H, W, debug=True)
    assert np.array_equal(gbin.cpu().numpy(), bin0)
    assert np.array_equal(glin.cpu().numpy(), lin0)
    # thousands of +-1 weights pile up on 6 pixels: tolerance scales with the count
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-3, atol=5e-3)
    assert float(got[1].abs().max()) == 0.0


def test_tiled_workspace_cleans_up_after_itself():
    """The control words of the tiled path (bucket cursors, overflow / finished
    counters) are zeroed by the kernels themselves: the workspace is
    zero-filled once (voxel._workspace) and every later call runs without a
    memset -- also after a call that overflowed its buckets."""
    from dvs_of_training_framework_amd import voxel
    B, C, H, W, n = 2, 5, 64, 96, 20000
    rng = np.random.default_rng(4)
    spread = synthetic.make_events(rng, B, H, W, n)
    crowded = {k: v.copy() for k, v in spread.items()}
    crowded['x'][:] = 7
    crowded['y'][:] = 9                      # every event in one pixel: overflow list
    t0 = torch.zeros(B, device='cuda')
    t1 = torch.full((B,), synthetic.WINDOW, device='cuda')
    z, w = np.zeros(B, np.float32), np.full(B, synthetic.WINDOW, np.float32)
    voxel._WORKSPACES.clear()
    for ev in (spread, crowded, spread, crowded, spread):
        want, _, lin0 = orc.voxelize(ev, z, w, B, C, H, W)
        got, _, glin = voxel.voxelize(dev_events(ev), t0, t1, B, C, H, W, debug=True)
        assert np.array_equal(glin.cpu().numpy(), lin0)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-3, atol=5e-3)
    assert len(voxel._WORKSPACES) == 1
    ws = next(iter(voxel._WORKSPACES.values()))
    control = voxel._lib.lib().dvsof_voxelize_control_bytes(B * n, B, C, H, W)
    assert control > 0 and int(ws[:control].to(torch.int32).abs().sum()) == 0


def test_polarity_is_a_sign_in_both_kernels():
    """VOXEL_SPEC: only the sign of the polarity counts (0 contributes nothing,
    |p| > 1 counts once) -- the thread-per-event kernel (n < 4096) and the
    tiled path agree with the oracle and so with each other."""
    from dvs_of_training_framework_amd.voxel import voxelize
    B, C, H, W = 2, 3, 32, 64
    z, w = np.zeros(B, np.float32), np.full(B, synthetic.WINDOW, np.float32)
    t0, t1 = torch.from_numpy(z).cuda(), torch.from_numpy(w).cuda()
    for n in (1000, 6000):                    # v1 kernel | tiled path
        rng = np.random.default_rng(n)
        ev = synthetic.make_events(rng, B, H, W, n)
        ev['polarity'] = rng.integers(-3, 4, B * n).astype(np.int64)   # includes 0, +-2, +-3
        want, bin0, lin0 = orc.voxelize(ev, z, w, B, C, H, W)
        got, gbin, glin = voxelize(dev_events(ev), t0, t1, B, C, H, W, debug=True)
        assert np.array_equal(gbin.cpu().numpy(), bin0)
        assert np.array_equal(glin.cpu().numpy(), lin0)
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=2e-5, rtol=1e-3)


def test_tiled_path_is_bitwise_reproducible_and_conserves_mass_exactly():
    """The tiled path accumulates in 2^-32 fixed point with integer LDS atomics:
    the result does not depend on the order of additions (two runs are bitwise
    equal, also with thousands of events on one pixel), and an event's two
    halves add up to exactly +-1, so the grid's total is the signed event count
    up to the final float rounding of each voxel."""
    from dvs_of_training_framework_amd.voxel import voxelize
    B, C, H, W, n = 2, 9, 96, 160, 40000
    rng = np.random.default_rng(77)
    ev = synthetic.make_events(rng, B, H, W, n)
    ev['x'][:5000] = 11
    ev['y'][:5000] = 13                      # a hot pixel
    t0 = torch.zeros(B, device='cuda')
    t1 = torch.full((B,), synthetic.WINDOW, device='cuda')
    a = voxelize(dev_events(ev), t0, t1, B, C, H, W)
    b = voxelize(dev_events(ev), t0, t1, B, C, H, W)
    assert torch.equal(a, b)
    want, _, _ = orc.voxelize(ev, np.zeros(B, np.float32), np.full(B, synthetic.WINDOW, np.float32),
                              B, C, H, W)
    np.testing.assert_allclose(a.cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    # per (sample, pixel): sum over bins == signed event count of that pixel
    per_pixel = a.sum(1).double().cpu().numpy()
    count = np.zeros((B, H, W))
    np.add.at(count, (ev['sample_index'], ev['y'], ev['x']), ev['polarity'])
    assert np.abs(per_pixel - count).max() <= 1e-3
