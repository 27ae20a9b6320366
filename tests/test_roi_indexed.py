"""-m gpu tests of the fast training-side RoiPool pair:

    mv3d_roi_pool_forward_views_indexed     pools every view AND builds the candidate index of the gradient in one launch
    mv3d_roi_pool_backward_views_indexed    zero fill + the ordered gather (roi_pooling_op.cc:319-452) in one launch

Bit-identical to the plain entries and to the oracle on BASELINE configs[2]'s full-size workload, on the pinned fixtures
(tests/golden/roipool_*), across channel widths, on a workspace that is reused call after call (the in-launch look-back cleans
up after itself), and loud -- NaN -- when handed an index that is not the views'.  Plus the call-compatible launcher aliases of
roi_pooling_op_gpu.h:18-27.  All calls go through the C-ABI (ctypes, mv3d_tf_amd.ops)."""
import ctypes as C

import numpy as np
import pytest

from mv3d_tf_amd import synth
from test_gpu_configs import VIEWS, dev, three_view_rois
from test_roipool_pin import HASHED, SMALL, check_outputs, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from mv3d_tf_amd import build, ops
    build.build()
    return torch, ops


@pytest.mark.parametrize("cold", [False, True])
def test_config2_indexed_pair_equals_plain_entries_and_oracle(gpu, oracle, cold):
    """BASELINE configs[2] at full size (3 maps x batch 2, R = 256 rows per view): forward + index in one launch, backward in one
    launch; top / argmax / bottom_diff equal to the oracle AND to the plain entries, five batches through ONE workspace."""
    torch, ops = gpu
    B, per = 2, 128
    maps = {k: synth.feature_map(170 + i, H, W, 512, B) for i, (k, (H, W)) in enumerate(VIEWS.items())}
    d_maps = {k: dev(torch, v) for k, v in maps.items()}
    ws = None
    for it in range(5):
        rois = three_view_rois(oracle, B, per if it != 3 else 37, 900 + 10 * it)          # (one batch with fewer rows)
        d_rois = {k: dev(torch, v) for k, v in rois.items()}
        outs, ws = ops.roi_pool_forward_views_indexed([(d_maps[k], d_rois[k], 0.125) for k in VIEWS], 7, 7, cold_maps=cold, index_ws=ws)
        plain = ops.roi_pool_forward_views([(d_maps[k], d_rois[k], 0.125) for k in VIEWS], 7, 7)
        grads = {}
        for k, (top, am), (ptop, pam) in zip(VIEWS, outs, plain):
            assert torch.equal(top, ptop) and torch.equal(am, pam), k
            if it < 2:
                o_top, o_am = oracle.roi_pool(maps[k], rois[k], 7, 7, 0.125)
                assert np.array_equal(top.cpu().numpy(), o_top) and np.array_equal(am.cpu().numpy(), o_am), k
            grads[k] = dev(torch, np.random.RandomState(31 + it).uniform(-1, 1, tuple(top.shape)).astype(np.float32))
        views = [(grads[k], d_rois[k], am, maps[k].shape, 0.125) for k, (_, am) in zip(VIEWS, outs)]
        got = ops.roi_pool_backward_views_indexed(views, 7, 7, ws)
        want = ops.roi_pool_backward_views(views, 7, 7)
        for k, a, b in zip(VIEWS, got, want):
            assert torch.equal(a, b), (it, k)
            if it < 2:
                o = oracle.roi_pool_grad(maps[k], rois[k], outs[list(VIEWS).index(k)][1].cpu().numpy(), grads[k].cpu().numpy(), 7, 7, 0.125)
                assert np.array_equal(a.cpu().numpy(), o), (it, k)
        # the look-back words and the done counter are zero again after every forward (the workspace's contract)
        head = ws[:256].view(torch.int32).cpu().numpy()
        assert head[0] == 0 and head[3] == 0 and head[2] != 0


@pytest.mark.parametrize("name", SMALL + HASHED)
def test_indexed_pair_on_the_pinned_fixtures(gpu, name):
    """the Appendix-D fixtures (real proposal boxes, the edge cases: out-of-map, 1x1, negative, +-.5 rounding, batch index > 0,
    ties, NaN, a stack of ROIs on one pixel) through the indexed pair; widths below 256 channels take the entries' plain path"""
    torch, ops = gpu
    g, data, rois, grad = load_case(name)
    d, r = dev(torch, data), dev(torch, rois)
    (res,), ws = ops.roi_pool_forward_views_indexed([(d, r, 0.125)], 7, 7)
    top, am = res
    bd, = ops.roi_pool_backward_views_indexed([(dev(torch, grad), r, am, data.shape, 0.125)], 7, 7, ws)
    check_outputs(g, top.cpu().numpy(), am.cpu().numpy(), bd.cpu().numpy())


@pytest.mark.parametrize("C", [64, 256, 320, 512, 1024])
@pytest.mark.parametrize("R", [40, 700])
def test_indexed_pair_other_widths_and_many_rois(gpu, oracle, C, R):
    """256 / 512 channels: the fused kernels (R = 700: three passes of the index's 256-ROI filter); 64 / 320 / 1024: the plain forward
    and the index-on-demand / sliced / generic backward behind the same two entries"""
    torch, ops = gpu
    rs = np.random.RandomState(C + R)
    B = 2
    maps = [rs.uniform(-1, 1, (B, 12, 20, C)).astype(np.float32), rs.uniform(-1, 1, (B, 9, 7, C)).astype(np.float32)]
    rois = []
    for m in maps:
        h, w = m.shape[1] * 8, m.shape[2] * 8
        x1, y1 = rs.randint(-8, w - 8, R), rs.randint(-8, h - 8, R)
        rois.append(np.stack([rs.randint(0, B, R), x1, y1, x1 + rs.randint(0, w // 2, R), y1 + rs.randint(0, h // 2, R)], 1).astype(np.float32))
    d_maps, d_rois = [dev(torch, m) for m in maps], [dev(torch, r) for r in rois]
    outs, ws = ops.roi_pool_forward_views_indexed([(m, r, 0.125) for m, r in zip(d_maps, d_rois)], 7, 7)
    grads = []
    for m, r, (top, am) in zip(maps, rois, outs):
        o_top, o_am = oracle.roi_pool(m, r, 7, 7, 0.125)
        assert np.array_equal(top.cpu().numpy(), o_top) and np.array_equal(am.cpu().numpy(), o_am)
        grads.append(rs.uniform(-1, 1, o_top.shape).astype(np.float32))
    bds = ops.roi_pool_backward_views_indexed([(dev(torch, g), r, am, m.shape, 0.125) for g, r, (_, am), m in zip(grads, d_rois, outs, maps)], 7, 7, ws)
    for m, r, (_, am), g, bd in zip(maps, rois, outs, grads, bds):
        assert np.array_equal(bd.cpu().numpy(), oracle.roi_pool_grad(m, r, am.cpu().numpy(), g, 7, 7, 0.125))


def test_an_index_of_other_views_is_refused_with_nan(gpu):
    torch, ops = gpu
    rs = np.random.RandomState(5)
    m = dev(torch, rs.uniform(-1, 1, (1, 10, 12, 256)).astype(np.float32))
    mk = lambda n: dev(torch, np.stack([np.zeros(n), rs.randint(0, 40, n), rs.randint(0, 30, n), rs.randint(40, 90, n), rs.randint(30, 70, n)], 1).astype(np.float32))
    r1, r2 = mk(20), mk(20)
    (res1,), ws = ops.roi_pool_forward_views_indexed([(m, r1, 0.125)], 7, 7)
    (res2,), ws2 = ops.roi_pool_forward_views_indexed([(m, r2, 0.125)], 7, 7)
    g = torch.ones_like(res1[0])
    ok, = ops.roi_pool_backward_views_indexed([(g, r1, res1[1], tuple(m.shape), 0.125)], 7, 7, ws)
    assert torch.isfinite(ok).all()
    bad, = ops.roi_pool_backward_views_indexed([(g, r2, res2[1], tuple(m.shape), 0.125)], 7, 7, ws)     # ws holds r1's index
    assert torch.isnan(bad).all()
    fresh = torch.zeros_like(ws)                                                                       # no forward at all
    bad2, = ops.roi_pool_backward_views_indexed([(g, r1, res1[1], tuple(m.shape), 0.125)], 7, 7, fresh)
    assert torch.isnan(bad2).all()


def test_autograd_views_function_uses_the_indexed_pair(gpu, oracle):
    torch, ops = gpu
    from mv3d_tf_amd.roi_pooling_layer.roi_pooling_op import roi_pool_views
    rs = np.random.RandomState(9)
    maps = [rs.uniform(-1, 1, (2, 12, 20, 512)).astype(np.float32), rs.uniform(-1, 1, (2, 9, 7, 512)).astype(np.float32)]
    rois = [np.stack([rs.randint(0, 2, 30), rs.randint(0, 60, 30), rs.randint(0, 40, 30), rs.randint(60, 150, 30), rs.randint(40, 90, 30)], 1).astype(np.float32)
            for _ in maps]
    xs = [dev(torch, m).requires_grad_(True) for m in maps]
    for rep in range(3):                                               # (the workspace goes back to the pool and is reused)
        for x in xs:
            x.grad = None
        tops = roi_pool_views([(x, dev(torch, r)) for x, r in zip(xs, rois)], 7, 7, 0.125)
        w = [dev(torch, rs.uniform(-1, 1, tuple(t.shape)).astype(np.float32)) for t in tops]
        sum((t * wi).sum() for t, wi in zip(tops, w)).backward()
        for m, r, x, wi in zip(maps, rois, xs, w):
            _, o_am = oracle.roi_pool(m, r, 7, 7, 0.125)
            assert np.array_equal(x.grad.cpu().numpy(), oracle.roi_pool_grad(m, r, o_am, wi.cpu().numpy(), 7, 7, 0.125))


def test_reference_launcher_aliases(gpu, oracle):
    """mv3d_ROIPoolForwardLaucher / mv3d_ROIPoolBackwardLaucher: roi_pooling_op_gpu.h:18-27's argument lists, through ctypes"""
    torch, ops = gpu
    from mv3d_tf_amd._lib import lib
    L = lib()
    rs = np.random.RandomState(3)
    data = rs.uniform(-1, 1, (2, 9, 11, 24)).astype(np.float32)
    rois = np.stack([rs.randint(0, 2, 12), rs.randint(0, 40, 12), rs.randint(0, 30, 12), rs.randint(40, 90, 12), rs.randint(30, 70, 12)], 1).astype(np.float32)
    d, r = dev(torch, data), dev(torch, rois)
    top = torch.empty((12, 7, 7, 24), device="cuda")
    am = torch.empty((12, 7, 7, 24), dtype=torch.int32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    assert L.mv3d_ROIPoolForwardLaucher(P(d), C.c_float(0.125), 12, 9, 11, 24, 7, 7, P(r), P(top), P(am), st) == 1
    o_top, o_am = oracle.roi_pool(data, rois, 7, 7, 0.125)
    assert np.array_equal(top.cpu().numpy(), o_top) and np.array_equal(am.cpu().numpy(), o_am)
    g = rs.uniform(-1, 1, o_top.shape).astype(np.float32)
    bd = torch.empty_like(d)
    assert L.mv3d_ROIPoolBackwardLaucher(P(dev(torch, g)), C.c_float(0.125), 2, 12, 9, 11, 24, 7, 7, P(r), P(bd), P(am), st) == 1
    assert np.array_equal(bd.cpu().numpy(), oracle.roi_pool_grad(data, rois, o_am, g, 7, 7, 0.125))
    assert L.mv3d_ROIPoolForwardLaucher(None, C.c_float(0.125), 12, 9, 11, 24, 7, 7, P(r), P(top), P(am), st) == 0       # refused, not exit(-1)
