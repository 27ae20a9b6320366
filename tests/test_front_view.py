"""Third (front-view) ROI, row X1.  PARITY UNPINNED: the reference has no front-view projection
(lib/networks/network.py:293-315 returns None), so the checks are properties of the documented definition
(mv3d_tf_amd/csrc/front_view.hip) plus device == oracle restatement, bit for bit."""
import numpy as np
import pytest

D_THETA = (np.pi / 2) / 512
D_PHI = np.radians(26.9) / 64


def boxes(seed, n):
    rng = np.random.RandomState(seed)
    r = np.zeros((n, 7), np.float32)
    r[:, 0] = rng.randint(0, 3, n)
    r[:, 1] = rng.uniform(2, 70, n); r[:, 2] = rng.uniform(-40, 40, n); r[:, 3] = rng.uniform(-2.5, 1.0, n)
    r[:, 4] = rng.uniform(0.5, 6, n); r[:, 5] = rng.uniform(0.4, 2.5, n); r[:, 6] = rng.uniform(0.5, 2.5, n)
    return r


def test_defined_atan2_matches_libm(oracle):
    rng = np.random.RandomState(3)
    y = np.concatenate([rng.normal(0, 20, 4000), [0, 0, 1, -1, 0.0, 5, -5, 1e-300, 1e300]])
    x = np.concatenate([rng.normal(0, 20, 4000), [1, -1, 0, 0, 0.0, 5, -5, 1e300, 1e-300]])
    assert np.abs(oracle.fv_atan2(y, x) - np.arctan2(y, x)).max() < 1e-15
    assert np.isnan(oracle.fv_atan2([np.nan], [1.0])[0])


def test_fv_box_contains_its_corners_and_is_monotone(oracle):
    r = boxes(5, 500)
    fv = oracle.rois_3d_to_fv(r)
    assert np.array_equal(fv[:, 0], r[:, 0])
    assert (fv[:, 1] <= fv[:, 3]).all() and (fv[:, 2] <= fv[:, 4]).all()
    assert fv[:, 1:].min() >= 0 and fv[:, [1, 3]].max() <= 511 and fv[:, [2, 4]].max() <= 63
    # the box contains the projection of its 8 corners (libm projection, +-1 cell for the floor at a cell edge)
    cn = oracle.lidar_3d_to_corners(r[:, 1:7]).astype(np.float64)
    x, y, z = cn[:, 0:8], cn[:, 8:16], cn[:, 16:24]
    col = np.clip(np.floor((np.pi / 4 - np.arctan2(y, x)) / D_THETA), 0, 511)
    row = np.clip(np.floor((np.radians(2.0) - np.arctan2(z, np.hypot(x, y))) / D_PHI), 0, 63)
    assert (col.min(1) >= fv[:, 1] - 1).all() and (col.max(1) <= fv[:, 3] + 1).all()
    assert (row.min(1) >= fv[:, 2] - 1).all() and (row.max(1) <= fv[:, 4] + 1).all()
    # monotone: moving a box to the right (smaller y = smaller azimuth) never moves its columns left;
    # lowering it (smaller z) never moves its rows up
    base = np.array([[0, 20, 10, -1, 4, 1.6, 1.5]], np.float32).repeat(40, 0)
    base[:, 2] = np.linspace(15, -15, 40)
    f = oracle.rois_3d_to_fv(base)
    assert (np.diff(f[:, 1]) >= 0).all() and (np.diff(f[:, 3]) >= 0).all()
    base[:, 2] = 0; base[:, 3] = np.linspace(0.5, -3, 40)
    f = oracle.rois_3d_to_fv(base)
    assert (np.diff(f[:, 2]) >= 0).all() and (np.diff(f[:, 4]) >= 0).all()
    # a box straight ahead at sensor height sits on the centre columns / top rows
    c = oracle.rois_3d_to_fv(np.array([[0, 30, 0, 0, 2, 2, 0.5]], np.float32))[0]
    assert c[1] < 256 <= c[3] and c[2] <= 4 + 1 and c[4] >= 4


@pytest.mark.gpu
def test_device_equals_oracle(oracle):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from mv3d_tf_amd import build
    build.build()
    from mv3d_tf_amd.utils import front_view
    r = boxes(9, 5000)
    r[:6, 1:] = [[np.nan, 0, 0, 1, 1, 1], [1e30, 0, 0, 1, 1, 1], [0, 0, 0, 0, 0, 0], [-5, 2, -1, 4, 1.6, 1.5],
                 [0.2, 0, -0.9, 3.9, 1.6, 1.5], [30, 0, -1, 200, 200, 3]]
    got = front_view.rois_3d_to_fv(r)
    assert np.array_equal(got, oracle.rois_3d_to_fv(r))
    t = front_view.rois_3d_to_fv(torch.as_tensor(r).cuda())
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), got)
    assert front_view.rois_3d_to_fv(np.zeros((0, 7), np.float32)).shape == (0, 5)
