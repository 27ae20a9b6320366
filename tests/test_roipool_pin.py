"""RoiPool / RoiPoolGrad parity pin (SURVEY.md §8(c) "ROI pool", Appendix D `roipool_*`).

There is no runnable reference for this op, so the pin is: two independent restatements (oracle/mv3d_oracle.c from
roi_pooling_op.cc, tests/roipool_restatement.py from roi_pooling_op_gpu.cu.cc) agree bit for bit on the committed
fixtures, the analytic gradient equals a finite-difference gradient on tie-free inputs, and the HIP kernels
reproduce the fixtures (the `gpu` tests below, through the C-ABI)."""
import numpy as np
import pytest

import roipool_restatement as rs
from conftest import golden
from mv3d_tf_amd import synth

SMALL = ["roipool_bev_R128", "roipool_bev_R300", "roipool_rgb_R128", "roipool_rgb_R300", "roipool_edge"]
HASHED = ["roipool_bev_C512", "roipool_rgb_C512"]


def load_case(name):
    g = golden(name)
    B, H, W, C = (int(v) for v in g["shape"])
    data = synth.feature_map(int(g["map_seed"]), H, W, C, B)
    if int(g["ties"]):
        data[0, 0, 0, :] = 1.5
        data[0, 0, 1 % W, :] = 1.5
        data[0, 2, 2, 0] = np.nan
    assert synth.sha256(data) == str(g["data_sha"])
    rois = g["rois"]
    grad = np.random.RandomState(int(g["grad_seed"])).uniform(-1, 1, (rois.shape[0], 7, 7, C)).astype(np.float32)
    return g, data, rois, grad


def check_outputs(g, top, am, bd):
    assert synth.sha256(np.ascontiguousarray(top)) == str(g["top_sha"])
    assert synth.sha256(np.ascontiguousarray(am)) == str(g["argmax_sha"])
    assert synth.sha256(np.ascontiguousarray(bd)) == str(g["bottom_diff_sha"])
    if "top" in g.files:
        assert np.array_equal(top, g["top"], equal_nan=True)
        assert np.array_equal(am, g["argmax"]) and np.array_equal(bd, g["bottom_diff"])


@pytest.mark.parametrize("name", SMALL + HASHED)
def test_oracle_matches_fixture(oracle, name):
    g, data, rois, grad = load_case(name)
    top, am = oracle.roi_pool(data, rois, 7, 7, 0.125)
    check_outputs(g, top, am, oracle.roi_pool_grad(data, rois, am, grad, 7, 7, 0.125))


@pytest.mark.parametrize("name", ["roipool_bev_R128", "roipool_rgb_R128", "roipool_edge"])
def test_independent_restatement_matches_fixture(name):
    g, data, rois, grad = load_case(name)
    top, am = rs.forward(data, rois, 7, 7, 0.125)
    B, H, W, _ = data.shape
    check_outputs(g, top, am, rs.backward(grad, am, rois, B, H, W, 7, 7, 0.125))


def tie_free_map(seed, B, H, W, C):
    """every value distinct, neighbours in value >= 1/64 apart"""
    rng = np.random.RandomState(seed)
    return (rng.permutation(B * H * W * C).astype(np.float32) / np.float32(64.0)).reshape(B, H, W, C)


FD_ROIS = np.array([[0, 8, 8, 60, 50], [1, 0, 0, 95, 79], [0, 16, 24, 16, 24], [0, 3, 5, 70, 33], [1, 40, 8, 90, 70],
                    [0, -20, -20, 30, 30]], np.float32)


def test_finite_difference_gradient_oracle_and_restatement(oracle):
    B, H, W, C = 2, 10, 12, 3
    data = tie_free_map(5, B, H, W, C)
    wt = np.random.RandomState(6).uniform(0.5, 1.5, (len(FD_ROIS), 7, 7, C)).astype(np.float32)
    top, am = oracle.roi_pool(data, FD_ROIS, 7, 7, 0.125)
    analytic = oracle.roi_pool_grad(data, FD_ROIS, am, wt, 7, 7, 0.125)
    assert np.array_equal(analytic, rs.backward(wt, am, FD_ROIS, B, H, W, 7, 7, 0.125))
    rng = np.random.RandomState(7)
    hot = np.argwhere(analytic != 0)
    cold = np.argwhere(analytic == 0)
    pos = [tuple(p) for p in hot[rng.permutation(len(hot))[:40]]] + [tuple(p) for p in cold[rng.permutation(len(cold))[:20]]]
    for fwd in (oracle.roi_pool, rs.forward):
        fd = rs.finite_difference_grad(fwd, data, FD_ROIS, wt, 7, 7, 0.125, 2.0 ** -8, pos)
        want = np.array([analytic[p] for p in pos], np.float64)
        assert np.allclose(fd, want, rtol=1e-4, atol=1e-4), np.abs(fd - want).max()   # f32 forward sums, tolerance 1e-4


def test_gradient_conservation_in_map_rois(oracle):
    """well-formed in-map ROIs: every non-empty bin's gradient lands on exactly one input element"""
    data = tie_free_map(8, 2, 10, 12, 4)
    rois = FD_ROIS[:5]
    top, am = oracle.roi_pool(data, rois, 7, 7, 0.125)
    g = np.random.RandomState(9).uniform(0.5, 1, top.shape).astype(np.float32)
    bd = oracle.roi_pool_grad(data, rois, am, g, 7, 7, 0.125)
    assert (am >= 0).all() and np.isclose(bd.sum(dtype=np.float64), g.sum(dtype=np.float64), rtol=1e-6)


# ------------------------------------------------------------------ the HIP kernels against the pinned fixtures
@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from mv3d_tf_amd import build, ops
    build.build()
    return torch, ops


def _dev(torch, a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL + HASHED)
def test_hip_matches_fixture(gpu, name):
    torch, ops = gpu
    g, data, rois, grad = load_case(name)
    top, am = ops.roi_pool_forward(_dev(torch, data), _dev(torch, rois), 7, 7, 0.125)
    bd = ops.roi_pool_backward(_dev(torch, grad), _dev(torch, rois), am, data.shape, 7, 7, 0.125)
    check_outputs(g, top.cpu().numpy(), am.cpu().numpy(), bd.cpu().numpy())


@pytest.mark.gpu
def test_hip_finite_difference_gradient(gpu):
    torch, ops = gpu
    B, H, W, C = 2, 10, 12, 4
    data = tie_free_map(15, B, H, W, C)
    wt = np.random.RandomState(16).uniform(0.5, 1.5, (len(FD_ROIS), 7, 7, C)).astype(np.float32)
    rois_d = _dev(torch, FD_ROIS)

    def fwd(d, rois, ph, pw, sc):
        t, a = ops.roi_pool_forward(_dev(torch, d), rois_d, ph, pw, sc)
        return t.cpu().numpy(), a.cpu().numpy()

    top, am = ops.roi_pool_forward(_dev(torch, data), rois_d, 7, 7, 0.125)
    analytic = ops.roi_pool_backward(_dev(torch, wt), rois_d, am, data.shape, 7, 7, 0.125).cpu().numpy()
    rng = np.random.RandomState(17)
    hot = np.argwhere(analytic != 0)
    pos = [tuple(p) for p in hot[rng.permutation(len(hot))[:32]]]
    fd = rs.finite_difference_grad(fwd, data, FD_ROIS, wt, 7, 7, 0.125, 2.0 ** -8, pos)
    assert np.allclose(fd, np.array([analytic[p] for p in pos], np.float64), rtol=1e-4, atol=1e-4)
