"""The numpy restatement of the reference's augmentation
(oracle.cpu_oracle.augment_sample) satisfies the reference's own property
tests (tests/dataset/test_dataset.py:76-218) on its HDF5 fixtures."""
import numpy as np

from oracle import cpu_oracle as orc

H, W = 260, 346


def elem(fixtures, i):
    ev = fixtures[f'events_{i}']
    return fixtures['frames'][i:i + 2], ev[:, 0].astype(int), ev[:, 1].astype(int)


def test_flip(fixtures):
    img, x, y = elem(fixtures, 1)
    a, ax, ay = orc.augment_sample(img, x, y, True, 0, (0, 0, H, W))
    b, bx, by = orc.augment_sample(img, x, y, False, 0, (0, 0, H, W))
    assert (a != b).any()
    assert all((a[i][ay, ax] == b[i][by, bx]).all() for i in range(2))


def test_rotation_90(fixtures):
    img, x, y = elem(fixtures, 1)
    r, rx, ry = orc.augment_sample(img, x, y, False, 90, (0, 0, H, W))
    o, _, _ = orc.augment_sample(img, x, y, False, 0, (0, 0, H, W))
    keep = rx >= 0
    x0 = -(ry[keep] - H // 2) + W // 2
    y0 = (rx[keep] - W // 2) + H // 2
    assert (y0 < H).all() and (y0 >= 0).all() and (x0 < W).all() and (x0 >= 0).all()
    assert all((o[i][y0, x0] == r[i][ry[keep], rx[keep]]).all() for i in range(2))


def test_crop(fixtures):
    img, x, y = elem(fixtures, 1)
    c, cx, cy = orc.augment_sample(img, x, y, False, 0, (1, 2, 100, 150))
    assert c.shape[-2:] == (100, 150) and (c == img[:, 1:101, 2:152]).all()
    m = (x >= 2) & (x < 152) & (y >= 1) & (y < 101)
    assert np.array_equal(cx >= 0, m)
    assert (cx[m] == x[m] - 2).all() and (cy[m] == y[m] - 1).all()


def test_random_params_ranges():
    from dvs_of_training_framework_amd.augment import random_params
    f, a, b = random_params(64, (260, 346), (256, 256), 30, np.random.default_rng(0))
    assert f.dtype == bool and 10 < f.sum() < 54
    assert (a >= -30).all() and (a < 30).all()
    # np.random.randint(x - y): corner in [0, x - y), utils/data.py:107-117
    assert (b[:, 0] >= 0).all() and (b[:, 0] < 4).all() and (b[:, 1] < 90).all()
    assert (b[:, 2] == 256).all() and (b[:, 3] == 256).all()
    _, _, b0 = random_params(4, (256, 256), (256, 256))
    assert (b0[:, :2] == 0).all()
