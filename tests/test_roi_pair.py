"""-m gpu tests of the fast training-side RoiPool pair:

    mv3d_roi_pool_forward_views_pair     pools every view in one launch, the argmax plane kept as private one-byte codes (16-bit for bins of
                                         more than 255 pixels); one workgroup of the launch plans the backward's work list (csrc/roi_grad_plan.h)
    mv3d_roi_pool_backward_views_pair    ONE launch of LDS map tiles over that list, the reference's summation order (roi_pooling_op.cc:319-452)

top / bottom_diff bit-identical to the plain entries and to the oracle (the pair's private argmax plane through
mv3d_roi_pool_argmax_decode) on BASELINE configs[2]'s full-size workload, on the pinned fixtures
(tests/golden/roipool_*), across channel widths, with and without a workspace argument (filled with garbage first), on the planned
mode's edge cases (more hot tiles than the cap, empty flags, ROIs the estimate does not follow, 16-bit codes inside one-pixel units) and on
the static-grid fallback.  Plus the call-compatible launcher aliases of roi_pooling_op_gpu.h:18-27.  All calls go through the C-ABI
(ctypes, mv3d_tf_amd.ops)."""
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


# RoiPoolGrad of the pair called with a workspace and without one: since round 6 the same ONE launch of map tiles either way
# (csrc/roi_grad_tiles.hip; the entry ignores the workspace) -- both call forms stay covered, on garbage-filled scratch memory
WS = pytest.mark.parametrize("no_ws", [False, True], ids=["workspace", "no-workspace"])


@WS
@pytest.mark.parametrize("cold", [False, True])
def test_config2_pair_equals_plain_entries_and_oracle(gpu, oracle, cold, no_ws):
    """BASELINE configs[2] at full size (3 maps x batch 2, R = 256 rows per view): top / argmax (decoded) / bottom_diff equal to the
    oracle AND to the plain entries, five batches through ONE workspace."""
    torch, ops = gpu
    B, per = 2, 128
    maps = {k: synth.feature_map(170 + i, H, W, 512, B) for i, (k, (H, W)) in enumerate(VIEWS.items())}
    d_maps = {k: dev(torch, v) for k, v in maps.items()}
    ws = None
    for it in range(5):
        rois = three_view_rois(oracle, B, per if it != 3 else 37, 900 + 10 * it)          # (one batch with fewer rows)
        d_rois = {k: dev(torch, v) for k, v in rois.items()}
        fv = [(d_maps[k], d_rois[k], 0.125) for k in VIEWS]
        outs = ops.roi_pool_forward_views_pair(fv, 7, 7, cold_maps=cold)
        dec = ops.roi_pool_argmax_decode(fv, outs, 7, 7)              # the pair's private 16-bit plane -> the reference's int32 plane
        plain = ops.roi_pool_forward_views(fv, 7, 7)
        grads = {}
        for k, (top, _), am, (ptop, pam) in zip(VIEWS, outs, dec, plain):
            assert torch.equal(top, ptop) and torch.equal(am, pam), k
            if it < 2:
                o_top, o_am = oracle.roi_pool(maps[k], rois[k], 7, 7, 0.125)
                assert np.array_equal(top.cpu().numpy(), o_top) and np.array_equal(am.cpu().numpy(), o_am), k
            grads[k] = dev(torch, np.random.RandomState(31 + it).uniform(-1, 1, tuple(top.shape)).astype(np.float32))
        if no_ws:
            ws = False
        elif ws is None:
            from mv3d_tf_amd._lib import RoiGradView, lib
            arr = (RoiGradView * 3)(*[RoiGradView(0, 0, 0, 0, 0.125, B, 2 * per, H, W, 512) for H, W in VIEWS.values()])
            ws = torch.randint(0, 255, (lib().mv3d_roi_pool_pair_workspace_bytes(3, arr, 7, 7),), dtype=torch.uint8, device="cuda")
        got = ops.roi_pool_backward_views_pair([(grads[k], d_rois[k], am, maps[k].shape, 0.125) for k, (_, am) in zip(VIEWS, outs)], 7, 7, workspace=ws)
        want = ops.roi_pool_backward_views([(grads[k], d_rois[k], pam, maps[k].shape, 0.125) for k, (_, pam) in zip(VIEWS, plain)], 7, 7)
        for k, a, b, (_, pam) in zip(VIEWS, got, want, plain):
            assert torch.equal(a, b), (it, k)
            if it < 2:
                o = oracle.roi_pool_grad(maps[k], rois[k], pam.cpu().numpy(), grads[k].cpu().numpy(), 7, 7, 0.125)
                assert np.array_equal(a.cpu().numpy(), o), (it, k)


@WS
@pytest.mark.parametrize("name", SMALL + HASHED)
def test_pair_on_the_pinned_fixtures(gpu, name, no_ws):
    """the Appendix-D fixtures (real proposal boxes, the edge cases: out-of-map, 1x1, negative, +-.5 rounding, batch index > 0,
    ties, NaN, a stack of ROIs on one pixel) through the pair; widths below 256 channels take the entries' plain path"""
    torch, ops = gpu
    g, data, rois, grad = load_case(name)
    d, r = dev(torch, data), dev(torch, rois)
    res = ops.roi_pool_forward_views_pair([(d, r, 0.125)], 7, 7)
    top, am = res[0]
    dec, = ops.roi_pool_argmax_decode([(d, r, 0.125)], res, 7, 7)
    bd, = ops.roi_pool_backward_views_pair([(dev(torch, grad), r, am, data.shape, 0.125)], 7, 7, workspace=False if no_ws else None)
    check_outputs(g, top.cpu().numpy(), dec.cpu().numpy(), bd.cpu().numpy())


@WS
@pytest.mark.parametrize("C", [64, 256, 320, 512, 1024])
@pytest.mark.parametrize("R", [40, 700])
def test_pair_other_widths_and_many_rois(gpu, oracle, C, R, no_ws):
    """256 / 512 channels: the pair's kernels (R = 700: three passes of the index's 256-ROI filter); 64 / 320 / 1024: the plain forward
    and the plain indexed / sliced / generic backward behind the same two entries"""
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
    fv = [(m, r, 0.125) for m, r in zip(d_maps, d_rois)]
    outs = ops.roi_pool_forward_views_pair(fv, 7, 7)
    dec = ops.roi_pool_argmax_decode(fv, outs, 7, 7)
    grads, ams = [], []
    for m, r, (top, _), am in zip(maps, rois, outs, dec):
        o_top, o_am = oracle.roi_pool(m, r, 7, 7, 0.125)
        assert np.array_equal(top.cpu().numpy(), o_top) and np.array_equal(am.cpu().numpy(), o_am)
        grads.append(rs.uniform(-1, 1, o_top.shape).astype(np.float32)); ams.append(o_am)
    bds = ops.roi_pool_backward_views_pair([(dev(torch, g), r, am, m.shape, 0.125) for g, r, (_, am), m in zip(grads, d_rois, outs, maps)], 7, 7,
                                           workspace=False if no_ws else None)
    for m, r, o_am, g, bd in zip(maps, rois, ams, grads, bds):
        assert np.array_equal(bd.cpu().numpy(), oracle.roi_pool_grad(m, r, o_am, g, 7, 7, 0.125))


@WS
def test_pair_huge_rois_ties_and_nan(gpu, oracle, no_ws):
    """ROIs larger than the map (one bin = up to the whole map: the largest scan positions a 16-bit code has to hold), ROIs of one
    pixel, ties (first maximum wins), NaN pixels (never win), a stack of identical ROIs on one spot, rows past the rounded end"""
    torch, ops = gpu
    rs = np.random.RandomState(77)
    B, H, W, C = 2, 46, 155, 256
    m = rs.uniform(-1, 1, (B, H, W, C)).astype(np.float32)
    m[0, 3:9, 10:30, :] = 0.75                                       # ties
    m[1, 20, 40, :7] = np.nan
    m[1, 21:23, 41:44, 5] = np.nan
    rois = [[0, -4000, -3000, 9000, 7000], [1, 0, 0, W * 8 - 1, H * 8 - 1], [0, 80, 24, 80, 24], [1, 300, 160, 330, 180],
            [1, 159.5, 83.5, 344.5, 183.5], [0, 1236, 364, 1300, 400], [1, -50, -50, 10, 10]] + [[0, 64, 32, 200, 120]] * 40
    for _ in range(150):
        x1, y1 = rs.randint(-40, W * 8), rs.randint(-40, H * 8)
        rois.append([rs.randint(0, B), x1, y1, x1 + rs.randint(0, 900), y1 + rs.randint(0, 300)])
    rois = np.asarray(rois, np.float32)
    d, r = dev(torch, m), dev(torch, rois)
    res = ops.roi_pool_forward_views_pair([(d, r, 0.125)], 7, 7)
    dec, = ops.roi_pool_argmax_decode([(d, r, 0.125)], res, 7, 7)
    o_top, o_am = oracle.roi_pool(m, rois, 7, 7, 0.125)
    assert np.array_equal(res[0][0].cpu().numpy(), o_top, equal_nan=True) and np.array_equal(dec.cpu().numpy(), o_am)
    g = rs.uniform(-1, 1, o_top.shape).astype(np.float32)
    bd, = ops.roi_pool_backward_views_pair([(dev(torch, g), r, res[0][1], m.shape, 0.125)], 7, 7, workspace=False if no_ws else None)
    assert np.array_equal(bd.cpu().numpy(), oracle.roi_pool_grad(m, rois, o_am, g, 7, 7, 0.125))


@WS
@pytest.mark.parametrize("C", [256, 512])
def test_pair_malformed_and_overhanging_rois(gpu, oracle, C, no_ws):
    """The reference's backward lets a gradient through only inside the rounded ROI (roi_pooling_op.cc:401-404).  (a) A ROI whose end lies
    before its start is pooled by the forward as a forced 1 x 1 region and gets NO gradient; (b) extents 57, 114, 121 (f32: 7 * (57 / 7) >
    57): the last bin of the forward reaches one column / row past the ROI, what lands there is dropped; (c) a ROI of more than 2048 map
    pixels (the tile kernel evaluates the reference's per-pixel expressions there).  Both RoiPoolGrad structures against the oracle."""
    torch, ops = gpu
    rs = np.random.RandomState(C)
    B, H, W = 2, 70, 130
    m = rs.uniform(-1, 1, (B, H, W, C)).astype(np.float32)
    # rising ramps along both axes: a bin's maximum sits in its LAST row / column, i.e. in the overhang when there is one
    m += (np.arange(W, dtype=np.float32)[None, None, :, None] + np.arange(H, dtype=np.float32)[None, :, None, None]) * np.float32(4)
    rois = [[0, 0, 0, 448, 100], [0, 100, 50, 40, 80], [1, 100, 50, 140, 20], [1, 40, 80, 30, 10],
            [0, 8, 8, 8 + 113 * 8, 8 + 56 * 8], [1, 16, 0, 16 + 120 * 8, 456], [0, 0, 16, 500, 16 + 56 * 8], [1, 24, 24, 24 + 56 * 8, 24 + 56 * 8],
            [0, -20000, -20000, 20000, 20000], [1, -100, -40000, 300, 40000]]
    for _ in range(60):
        x1, y1 = rs.randint(-40, W * 8), rs.randint(-40, H * 8)
        rois.append([rs.randint(0, B), x1, y1, x1 + rs.choice([56 * 8, 113 * 8, 120 * 8, -30, 200]), y1 + rs.choice([56 * 8, -20, 90])])
    rois = np.asarray(rois, np.float32)
    d, r = dev(torch, m), dev(torch, rois)
    # (the pair's planes land on garbage, not on a fresh zero page: a big bin leaves its one-byte codes unwritten, a small one its 16-bit
    # codes -- a kernel that reads the wrong one must not get away with it)
    junk = torch.randint(0, 255, (192 << 20,), dtype=torch.uint8, device="cuda")
    del junk
    res = ops.roi_pool_forward_views_pair([(d, r, 0.125)], 7, 7)
    dec, = ops.roi_pool_argmax_decode([(d, r, 0.125)], res, 7, 7)
    o_top, o_am = oracle.roi_pool(m, rois, 7, 7, 0.125)
    assert np.array_equal(res[0][0].cpu().numpy(), o_top) and np.array_equal(dec.cpu().numpy(), o_am)
    # the case exists in this input: some argmax of ROI 0 names column 57 (past its end, 56)
    assert ((o_am[0] // C) % W == 57).any()
    g = rs.uniform(-1, 1, o_top.shape).astype(np.float32)
    want = oracle.roi_pool_grad(m, rois, o_am, g, 7, 7, 0.125)
    bd, = ops.roi_pool_backward_views_pair([(dev(torch, g), r, res[0][1], m.shape, 0.125)], 7, 7, workspace=False if no_ws else None)
    assert np.array_equal(bd.cpu().numpy(), want)


def _pair_against_oracle(torch, ops, oracle, m, rois, seed):
    d, r = dev(torch, m), dev(torch, rois)
    junk = torch.randint(0, 255, (64 << 20,), dtype=torch.uint8, device="cuda")      # (the pair's planes and work list land on garbage)
    del junk
    res = ops.roi_pool_forward_views_pair([(d, r, 0.125)], 7, 7)
    dec, = ops.roi_pool_argmax_decode([(d, r, 0.125)], res, 7, 7)
    o_top, o_am = oracle.roi_pool(m, rois, 7, 7, 0.125)
    assert np.array_equal(res[0][0].cpu().numpy(), o_top) and np.array_equal(dec.cpu().numpy(), o_am)
    g = np.random.RandomState(seed).uniform(-1, 1, o_top.shape).astype(np.float32)
    want = oracle.roi_pool_grad(m, rois, o_am, g, 7, 7, 0.125)
    for _ in range(2):                                               # (the list is read again by a second backward of the same forward)
        bd, = ops.roi_pool_backward_views_pair([(dev(torch, g), r, res[0][1], m.shape, 0.125)], 7, 7, workspace=False)
        assert np.array_equal(bd.cpu().numpy(), want)
    return res


def _work_list(torch, res, m, R, C):
    """the backward's work list as the forward's planning workgroup wrote it (csrc/roi_grad_plan.h): last quarter of view 0's argmax buffer"""
    raw = res[0][1].view(torch.uint8).reshape(-1)[3 * R * 49 * C:].cpu().numpy()
    n = int(raw[:4].view(np.int32)[0])
    return raw[256:256 + 16 * n].view(np.int32).reshape(-1, 4)


def _assert_partition(units, B, H, W):
    """every pixel of every frame belongs to exactly one unit that is not skipped (a sub-tile outside the map), estimates never rise along
    the whole tiles of the list"""
    cover = np.zeros((B, H, W), np.int32)
    for x, y0, x0, _ in units:
        if x & (1 << 24):
            continue
        b, th, tw = (x >> 4) & 0xfff, 1 << ((x >> 16) & 15), 1 << ((x >> 20) & 15)
        cover[b, y0:y0 + th, x0:x0 + tw] += 1
    assert (cover == 1).all()


def test_planned_backward_more_hot_tiles_than_the_cap(gpu, oracle):
    """The pair's forward launch plans its backward (csrc/roi_grad_plan.h): tiles under long entry streams are cut into four sub-tiles, at most
    RGT_HOT_MAX = 128 of them.  A dense view where EVERY tile is hot (256 ROIs, each over one half of a 16 x 64 map: 256 tiles of 2 x 2): 128 tiles cut into
    one-pixel units (register sums), the others whole, no unit flagged empty; bottom_diff equal to the oracle."""
    torch, ops = gpu
    rs = np.random.RandomState(5)
    B, H, W, C, R = 1, 16, 64, 256, 256
    m = rs.uniform(-1, 1, (B, H, W, C)).astype(np.float32)
    # every ROI covers one half of the map (8 x 16 = 128 tiles: the most the estimate follows), left and right halves alternating
    side = np.arange(R) % 2
    x1 = side * 256 + rs.randint(0, 8, R)
    x2 = np.where(side == 0, 247 - rs.randint(0, 8, R), 511)
    rois = np.stack([np.zeros(R), x1, rs.randint(0, 8, R), x2, 127 - rs.randint(0, 4, R)], 1).astype(np.float32)
    res = _pair_against_oracle(torch, ops, oracle, m, rois, 6)
    units = _work_list(torch, res, m, R, C)
    shapes = (units[:, 0] >> 16) & 0xff                                # ths | tws << 4
    assert (shapes == 0).sum() == 4 * 128 and (shapes == 0x11).sum() == 256 - 128 and len(units) == 4 * 128 + 128
    assert not (units[:, 0] & (1 << 25)).any()
    _assert_partition(units, B, H, W)


def test_planned_backward_empty_tiles_big_rois_and_16_bit_codes_in_a_hot_pixel(gpu, oracle):
    """(a) tiles no ROI touches are flagged in the work list and only written as zeros; (b) ROIs over more than 128 tiles are not followed
    by the estimate: their view keeps every tile unflagged; (c) one-pixel units (register sums) that meet bins of more than 255 pixels
    (16-bit codes: the sum goes through its LDS slot and back)."""
    torch, ops = gpu
    rs = np.random.RandomState(9)
    B, H, W, C = 1, 100, 112, 256
    m = rs.uniform(-1, 1, (B, H, W, C)).astype(np.float32)
    hot = [[0, 8 * 30 + rs.randint(0, 8), 8 * 40 + rs.randint(0, 8), 8 * 36 + rs.randint(0, 16), 8 * 45 + rs.randint(0, 16)] for _ in range(380)]
    # (a): only the small stacked ROIs -- most of the map's 2 x 2 tiles are empty, the stack's are hot
    rois = np.asarray(hot, np.float32)
    res = _pair_against_oracle(torch, ops, oracle, m, rois, 10)
    units = _work_list(torch, res, m, len(rois), C)
    assert (units[:, 0] & (1 << 25)).sum() > 2000 and (((units[:, 0] >> 16) & 0xff) == 0).sum() >= 4
    _assert_partition(units, B, H, W)
    whole = units[((units[:, 0] >> 16) & 0xff) != 0]
    e = whole[:, 3]                                                  # longest estimate first: the classes (>= 250, >= 125, >= 31, > 0, 0) in turn
    cls = np.where(e >= 250, 0, np.where(e >= 125, 1, np.where(e >= 31, 2, np.where(e > 0, 3, 4))))
    assert (np.diff(cls) >= 0).all() and ((units[:, 0] & (1 << 25)) != 0)[((units[:, 0] >> 16) & 0xff) != 0][cls == 4].all()
    # (b) + (c): whole-map ROIs on top (bins of 16 x 16 = 256 pixels), interleaved with the stack
    rois = np.asarray(hot[:190] + [[0, 0, 0, W * 8 - 1, H * 8 - 1]] * 6 + hot[190:] + [[0, -40, -40, W * 8 + 40, H * 8 + 40]] * 4, np.float32)
    res = _pair_against_oracle(torch, ops, oracle, m, rois, 11)
    units = _work_list(torch, res, m, len(rois), C)
    assert not (units[:, 0] & (1 << 25)).any() and (((units[:, 0] >> 16) & 0xff) == 0).sum() >= 4


def test_backward_static_grid_when_the_planner_does_not_apply(gpu, oracle):
    """More tiles than the planner's heat map holds (a 250 x 250 map in 4 x 4 tiles: 3969 > 3072), or a work list larger than the unused
    quarter of view 0's argmax buffer: both launches decide for the static grid from the shapes alone; same results."""
    torch, ops = gpu
    rs = np.random.RandomState(12)
    B, H, W, C, R = 1, 250, 250, 256, 24
    m = rs.uniform(-1, 1, (B, H, W, C)).astype(np.float32)
    x1, y1 = rs.randint(-40, W * 8 - 200, R), rs.randint(-40, H * 8 - 200, R)
    rois = np.stack([np.zeros(R), x1, y1, x1 + rs.randint(0, 900, R), y1 + rs.randint(0, 900, R)], 1).astype(np.float32)
    _pair_against_oracle(torch, ops, oracle, m, rois, 13)
    # and a list that does not fit the unused quarter of the argmax buffer (ONE ROI on a 100 x 100 map: 625 tiles, 12.5 KB of codes)
    m = rs.uniform(-1, 1, (1, 100, 100, C)).astype(np.float32)
    _pair_against_oracle(torch, ops, oracle, m, np.asarray([[0, 100, 60, 420, 333]], np.float32), 14)


def test_autograd_views_function_uses_the_pair(gpu, oracle):
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


@pytest.mark.parametrize("R", [150, 2000])
@pytest.mark.parametrize("cold", [False, True])
def test_forward_views_without_argmax_is_the_same_top(gpu, oracle, R, cold):
    """inference (argmax_data = NULL in every view: the maximum-only scan of mv3d_roi_pool_forward_views[_cold]) gives the bits of the
    full op's `top` -- with -0.0 / +0.0 ties (the first one met stays), NaN pixels (never win), ROIs larger than the map, empty bins;
    R = 2000 rows per view takes the four-pass workgroups"""
    torch, ops = gpu
    rs = np.random.RandomState(R)
    B = 2
    maps = [rs.uniform(-1, 1, (B, 20, 31, 512)).astype(np.float32), rs.uniform(-1, 1, (B, 9, 40, 512)).astype(np.float32)]
    maps[0][0, 2:6, 3:9, :] = 0.0
    maps[0][0, 2:6, 3:9, ::2] = -0.0                                   # signed zeros under the pooling windows
    maps[0][1, 5, 7, :9] = np.nan
    maps[1][maps[1] < -0.2] = -0.0
    rois = []
    for m in maps:
        h, w = m.shape[1] * 8, m.shape[2] * 8
        x1, y1 = rs.randint(-30, w, R), rs.randint(-30, h, R)
        r = np.stack([rs.randint(0, B, R), x1, y1, x1 + rs.randint(0, w, R), y1 + rs.randint(0, h, R)], 1).astype(np.float32)
        r[:5] = [[0, -4000, -3000, 9000, 7000], [1, 0, 0, w - 1, h - 1], [0, 24, 16, 24, 16], [1, 9999, 9999, 10005, 10005], [0, 16, 16, 71, 47]]
        rois.append(r)
    fv = [(dev(torch, m), dev(torch, r), 0.125) for m, r in zip(maps, rois)]
    full = ops.roi_pool_forward_views(fv, 7, 7, cold_maps=cold)
    none = ops.roi_pool_forward_views(fv, 7, 7, cold_maps=cold, want_argmax=False)
    for m, r, (top, am), (top2, am2) in zip(maps, rois, full, none):
        assert am2 is None
        a, b = top.cpu().numpy(), top2.cpu().numpy()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))    # bit for bit, signs of zero included
        if R == 150:
            o_top, _ = oracle.roi_pool(m, r, 7, 7, 0.125)
            assert np.array_equal(b.view(np.uint32), o_top.view(np.uint32))


@pytest.mark.parametrize("half", ["float16", "bfloat16"])
@pytest.mark.parametrize("R,cold", [(150, False), (2000, True)])
def test_forward_views_with_16_bit_tops(gpu, half, R, cold):
    """mv3d_roi_pool_forward_views_half (the serving graph in 16-bit mode): the bits of the f32 `top` cast to that type, incl. NaN-free
    maxima of NaN pixels, signed zeros and empty bins; argmax planes or other channel widths are refused"""
    torch, ops = gpu
    T = getattr(torch, half)
    rs = np.random.RandomState(R + 1)
    B = 2
    maps = [rs.uniform(-3, 3, (B, 20, 31, 512)).astype(np.float32), rs.uniform(-1e-5, 1e5, (B, 9, 40, 512)).astype(np.float32)]
    maps[0][0, 2:6, 3:9, ::2] = -0.0
    maps[0][1, 5, 7, :9] = np.nan
    rois = []
    for m in maps:
        h, w = m.shape[1] * 8, m.shape[2] * 8
        x1, y1 = rs.randint(-30, w, R), rs.randint(-30, h, R)
        rois.append(np.stack([rs.randint(0, B, R), x1, y1, x1 + rs.randint(0, w, R), y1 + rs.randint(0, h, R)], 1).astype(np.float32))
    fv = [(dev(torch, m), dev(torch, r), 0.125) for m, r in zip(maps, rois)]
    full = ops.roi_pool_forward_views(fv, 7, 7, cold_maps=cold, want_argmax=False)
    got = ops.roi_pool_forward_views(fv, 7, 7, cold_maps=cold, want_argmax=False, top_dtype=T)
    for (top, _), (top16, am) in zip(full, got):
        assert am is None and top16.dtype == T
        assert torch.equal(top.to(T).view(torch.int16), top16.view(torch.int16))
    with pytest.raises(Exception):
        ops.roi_pool_forward_views([(dev(torch, maps[0][..., :64].copy()), fv[0][1], 0.125)], 7, 7, want_argmax=False, top_dtype=T)
