"""-m gpu tests for the BASELINE.json configurations the per-function tests do not reach at full size:

  configs[2]  full 3-view path: BEV 76x76x512 + RGB 46x155x512 + FV 8x64x512 feature maps, batch 2,
              RoiPool forward + backward of every view, against the oracle;
  configs[4]  per-GPU inference workload: batch 16, TEST cfg (6000 -> 300), the step captured in a hipGraph and
              replayed: replay == eager == oracle, frame by frame;
  a11 edges   the `project_edge` boxes (NaN, 1e30, behind the camera, 200 m) through the DEVICE projection
              (mv3d_proposal_target_stage2's rois_img) against the reference-generated golden.
All calls go through the C-ABI (ctypes, mv3d_tf_amd.ops)."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden
from mv3d_tf_amd import synth

pytestmark = pytest.mark.gpu

TEST_CFG = dict(RPN_PRE_NMS_TOP_N=6000, RPN_POST_NMS_TOP_N=300, RPN_NMS_THRESH=0.7, RPN_MIN_SIZE=5)
TRAIN_CFG = dict(RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000, RPN_NMS_THRESH=0.7, RPN_MIN_SIZE=5)
VIEWS = {"bev": (76, 76), "rgb": (46, 155), "fv": (8, 64)}


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from mv3d_tf_amd import build, ops
    build.build()
    return torch, ops


def dev(torch, a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda()


def three_view_rois(oracle, B, per_frame, seed0):
    """per-frame sampled ROIs of a batch: BEV / image boxes of real proposals, FV boxes from the cylindrical projection
    of the same 3D proposals; column 0 = frame index"""
    from mv3d_tf_amd.utils import front_view
    bev, rgb, fv = [], [], []
    for b in range(B):
        prob, pred, info, calib = synth.rpn_head(seed0 + b, 76, 76, "peaky")
        bv, img, b3 = oracle.proposal_layer_3d(prob, pred, info, calib, "TRAIN", [8, ], cfg={"TRAIN": TRAIN_CFG})
        for dst, src in ((bev, bv), (rgb, img)):
            r = src[:per_frame].copy(); r[:, 0] = b; dst.append(r)
        r3 = b3[:per_frame].copy(); r3[:, 0] = b
        f = front_view.rois_3d_to_fv(r3)                              # device kernel ...
        assert np.array_equal(f, oracle.rois_3d_to_fv(r3))            # ... equal to the oracle's restatement
        fv.append(f)
    return {"bev": np.concatenate(bev), "rgb": np.concatenate(rgb), "fv": np.concatenate(fv)}


def test_config2_three_views_batch2_forward_backward(gpu, oracle):
    """BASELINE configs[2] RoiPool workload at full size: 3 maps x batch 2, 128 sampled ROIs per frame (R = 256 per view),
    one forward launch for all views, one backward launch for all views; equal to the oracle bit for bit."""
    torch, ops = gpu
    B, per = 2, 128
    rois = three_view_rois(oracle, B, per, 700)
    maps = {k: synth.feature_map(70 + i, H, W, 512, B) for i, (k, (H, W)) in enumerate(VIEWS.items())}
    d_maps = {k: dev(torch, v) for k, v in maps.items()}
    d_rois = {k: dev(torch, v) for k, v in rois.items()}
    outs = ops.roi_pool_forward_views([(d_maps[k], d_rois[k], 0.125) for k in VIEWS], 7, 7)
    grads, want_am = {}, {}
    for (k, (top, am)) in zip(VIEWS, outs):
        o_top, o_am = oracle.roi_pool(maps[k], rois[k], 7, 7, 0.125)
        assert np.array_equal(top.cpu().numpy(), o_top), k
        assert np.array_equal(am.cpu().numpy(), o_am), k
        want_am[k] = o_am
        grads[k] = np.random.RandomState({"bev": 11, "rgb": 12, "fv": 13}[k]).uniform(-1, 1, o_top.shape).astype(np.float32)
    bds = ops.roi_pool_backward_views([(dev(torch, grads[k]), d_rois[k], am, maps[k].shape, 0.125)
                                       for k, (_, am) in zip(VIEWS, outs)], 7, 7)
    for k, bd in zip(VIEWS, bds):
        want = oracle.roi_pool_grad(maps[k], rois[k], want_am[k], grads[k], 7, 7, 0.125)
        assert np.array_equal(bd.cpu().numpy(), want), k
        # the single-view entry gives the same bytes
        one = ops.roi_pool_backward(dev(torch, grads[k]), d_rois[k], dev(torch, want_am[k]), maps[k].shape, 7, 7, 0.125)
        assert np.array_equal(one.cpu().numpy(), want), k


@pytest.mark.parametrize("C", [64, 128, 256, 320, 1024])
def test_views_entries_other_channel_widths(gpu, oracle, C):
    """The several-views entries on widths other than 512: 64 / 128 / 256 take the indexed RoiPoolGrad (1, 2, 4 channel
    slices), 1024 the sliced kernel behind the same call (the workspace is ignored), 320 the generic kernels; the cold-map
    forward gives the bytes of the plain one."""
    torch, ops = gpu
    rs = np.random.RandomState(C)
    B, H, W, R = 2, 12, 20, 40
    maps = [rs.uniform(-1, 1, (B, H, W, C)).astype(np.float32), rs.uniform(-1, 1, (B, 9, 7, C)).astype(np.float32)]
    rois = []
    for m in maps:
        h, w = m.shape[1] * 8, m.shape[2] * 8
        x1, y1 = rs.randint(-8, w - 8, R), rs.randint(-8, h - 8, R)
        rois.append(np.stack([rs.randint(0, B, R), x1, y1, x1 + rs.randint(0, w // 2, R), y1 + rs.randint(0, h // 2, R)], 1).astype(np.float32))
    d_maps, d_rois = [dev(torch, m) for m in maps], [dev(torch, r) for r in rois]
    outs = ops.roi_pool_forward_views([(m, r, 0.125) for m, r in zip(d_maps, d_rois)], 7, 7)
    cold = ops.roi_pool_forward_views([(m, r, 0.125) for m, r in zip(d_maps, d_rois)], 7, 7, cold_maps=True)
    grads, ams = [], []
    for m, r, (top, am), (ctop, cam) in zip(maps, rois, outs, cold):
        o_top, o_am = oracle.roi_pool(m, r, 7, 7, 0.125)
        assert np.array_equal(top.cpu().numpy(), o_top) and np.array_equal(am.cpu().numpy(), o_am)
        assert np.array_equal(ctop.cpu().numpy(), o_top) and np.array_equal(cam.cpu().numpy(), o_am)
        grads.append(rs.uniform(-1, 1, o_top.shape).astype(np.float32)); ams.append(o_am)
    bds = ops.roi_pool_backward_views([(dev(torch, g), r, am, m.shape, 0.125) for g, r, (_, am), m in zip(grads, d_rois, outs, maps)], 7, 7)
    for m, r, am, g, bd in zip(maps, rois, ams, grads, bds):
        assert np.array_equal(bd.cpu().numpy(), oracle.roi_pool_grad(m, r, am, g, 7, 7, 0.125))


def test_config4_batch16_hipgraph_replay_equals_eager_equals_oracle(gpu, oracle):
    """BASELINE configs[4] per-GPU step: 16 frames, TEST cfg, proposal_3d + FV ROIs + all THREE RoiPool views.  The captured hipGraph,
    replayed twice (the second time on fresh inputs written into the same buffers), gives the eager results and the
    oracle's, frame by frame."""
    torch, ops = gpu
    B = 16
    params = ops.proposal_params(TEST_CFG)

    def frames(seed0):
        hs = [synth.rpn_head(seed0 + b, 76, 76, "peaky" if b % 2 == 0 else "rand") for b in range(B)]
        return (np.concatenate([h[0] for h in hs]), np.concatenate([h[1] for h in hs]), np.concatenate([h[2] for h in hs]),
                np.stack([h[3] for h in hs]))

    host = frames(2000)
    prob, pred, info, calib = (dev(torch, a) for a in host)
    bev_h, rgb_h, fv_h = synth.feature_map(81, 76, 76, 512, B), synth.feature_map(82, 46, 155, 512, B), synth.feature_map(83, 8, 64, 512, B)
    bev, rgb, fvm = dev(torch, bev_h), dev(torch, rgb_h), dev(torch, fv_h)
    eager = ops.proposal_3d(prob, pred, info, calib, params)
    cap = eager[0].shape[1]
    e_fv = ops.rois_3d_to_fv(eager[2].view(-1, 7))                  # the third view's ROIs (zero rows -> zero boxes)

    def step(o, fv_rois, outs=None):
        ops.rois_3d_to_fv(o[2].view(-1, 7), out=fv_rois)
        return ops.roi_pool_forward_views([(bev, o[0].view(-1, 5), 0.125), (rgb, o[1].view(-1, 5), 0.125), (fvm, fv_rois, 0.125)], 7, 7, outs=outs)

    e_views = step(eager, e_fv)
    torch.cuda.synchronize()
    e_host = [t.cpu().numpy().copy() for t in eager] + [t.cpu().numpy().copy() for pair in e_views for t in pair] + [e_fv.cpu().numpy().copy()]

    out = tuple(torch.empty_like(t) for t in eager)
    fv_out = torch.empty_like(e_fv)
    v_out = [(torch.empty_like(t), torch.empty_like(a)) for (t, a) in e_views]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                              # warm-up on the capture stream (workspace allocation)
        ops.proposal_3d(prob, pred, info, calib, params, out=out)
        step(out, fv_out, v_out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.proposal_3d(prob, pred, info, calib, params, out=out)
        step(out, fv_out, v_out)
    for t in out:
        t.zero_()
    for rep in range(2):
        graph.replay()
        torch.cuda.synchronize()
        got = [t.cpu().numpy() for t in out] + [t.cpu().numpy() for pair in v_out for t in pair] + [fv_out.cpu().numpy()]
        for a, b in zip(got, e_host):
            assert np.array_equal(a, b), "replay %d differs from the eager step" % rep
    # oracle, frame by frame (every frame must reach 300 rows or fewer; rows past num_out are zero)
    num = out[3].cpu().numpy()
    for b in range(B):
        bv, img, b3 = oracle.proposal_layer_3d(host[0][b:b + 1], host[1][b:b + 1], host[2][b:b + 1], host[3][b], "TEST", [8, ],
                                               cfg={"TEST": TEST_CFG})
        n = bv.shape[0]
        assert num[b] == n and n <= cap
        bv[:, 0] = b; img[:, 0] = b; b3[:, 0] = b
        assert np.array_equal(got[0][b, :n], bv) and np.array_equal(got[1][b, :n], img) and np.array_equal(got[2][b, :n], b3)
        assert not got[0][b, n:].any()
    for b in (0, B - 1):                                        # RoiPool rows of the first and last frame vs the oracle
        rows = slice(b * cap, b * cap + int(num[b]))
        fv_rows = oracle.rois_3d_to_fv(got[2].reshape(-1, 7)[rows])
        assert np.array_equal(got[11][rows], fv_rows)
        for k, (fmap, blob) in enumerate(((bev_h, got[0].reshape(-1, 5)[rows]), (rgb_h, got[1].reshape(-1, 5)[rows]), (fv_h, fv_rows))):
            o_top, o_am = oracle.roi_pool(fmap, blob, 7, 7, 0.125)
            assert np.array_equal(got[5 + 2 * k][rows], o_top) and np.array_equal(got[6 + 2 * k][rows], o_am)
    # new inputs written into the captured buffers: the replay follows them
    host2 = frames(3000)
    for t, a in zip((prob, pred, info, calib), host2):
        t.copy_(torch.as_tensor(a))
    graph.replay()
    torch.cuda.synchronize()
    bv, img, b3 = oracle.proposal_layer_3d(host2[0][3:4], host2[1][3:4], host2[2][3:4], host2[3][3], "TEST", [8, ], cfg={"TEST": TEST_CFG})
    n = bv.shape[0]
    assert int(out[3][3].item()) == n and np.array_equal(out[0][3, :n, 1:].cpu().numpy(), bv[:, 1:])


def test_project_edge_boxes_through_device_projection(gpu):
    """a11 edge set on the device: the golden `img` of tests/golden/project_edge.npz (generated by the reference's
    lidar_cnr_to_img: NaN / 1e30 / depth <= 0 / 200 m boxes -> INT32_MIN and overflowed values) must come out of
    image_box() as used by mv3d_proposal_target_stage2 (rois_img = projected sampled 3D ROIs)."""
    torch, ops = gpu
    from mv3d_tf_amd._lib import ProposalTargetParams
    g = golden("project_edge")
    n = g["boxes3d"].shape[0]
    rois_3d = np.zeros((n, 7), np.float32); rois_3d[:, 1:] = g["boxes3d"]
    rois_bv = np.zeros((n, 5), np.float32); rois_bv[:, 1:] = [400, 400, 420, 440]          # IoU 0 with the GT below
    gt_bv = np.array([[100, 100, 116, 139, 1]], np.float32)
    gt_3d = np.array([[30.0, 10.0, -0.95, 3.9, 1.6, 1.56, 1]], np.float32)
    gt_cnr = np.concatenate([np.arange(24, dtype=np.float32), [1]])[None]
    params = ProposalTargetParams(2, 0, 0.5, 0.5, 0.0)                                       # bg = [0, 0.5): every edge roi
    d = lambda a: dev(torch, a)
    counts, ws = ops.proposal_target_stage1(d(rois_bv), d(rois_3d), d(gt_bv), d(gt_3d), params)
    ncand, n_fg, n_bg, _ = (int(v) for v in counts.cpu().numpy())
    assert (ncand, n_fg, n_bg) == (n + 1, 1, n)
    out = ops.proposal_target_stage2(d(rois_bv), d(rois_3d), d(gt_bv), d(gt_3d), d(gt_cnr), d(g["calib"]), params,
                                     np.arange(1), np.arange(n), ws)
    rois_img = out[1].cpu().numpy()
    assert rois_img.shape == (n + 1, 5)
    assert np.array_equal(rois_img[1:, 1:], g["img"].astype(np.float32))                     # int32 -> f32 like :88-92
    assert np.array_equal(out[4].cpu().numpy()[1:, 1:], g["boxes3d"], equal_nan=True)


@pytest.mark.parametrize("yml", [False, True])
def test_config2_train_path_batch_replay_vs_oracle(gpu, oracle, yml):
    """BASELINE configs[2] path-only step as bench.py runs it (mv3d_tf_amd.hot_path.TrainPathBatch): batch 2, TRAIN cfg
    12000 -> 2000, anchor targets, <= 128 sampled ROIs per frame, FV ROIs, RoiPool forward + backward on three views.
    setup() (host RNG in the loop) and the sync-free replay give identical bytes, equal to the oracle run frame by
    frame with the same numpy seed (draw for draw), the ROI batch column being the frame index."""
    torch, ops = gpu
    from mv3d_tf_amd import hot_path
    from mv3d_tf_amd.fast_rcnn.config import cfg
    B = 2
    frames = [synth.rpn_head(900 + b, 76, 76, "peaky", return_gt=True) for b in range(B)]
    maps = hot_path.synth_maps(B, 5, torch.device("cuda"))
    saved = {k: cfg.TRAIN[k] for k in ("BG_THRESH_LO", "BG_THRESH_HI", "FG_THRESH")}
    o_train = dict(oracle.TRAIN)
    if yml:                                                     # experiments/cfgs/faster_rcnn_end2end.yml:10-12 (bench.py's setting)
        o_train.update(BG_THRESH_LO=0.0, BG_THRESH_HI=0.5, FG_THRESH=0.7)
        cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.BG_THRESH_HI, cfg.TRAIN.FG_THRESH = 0.0, 0.5, 0.7
    try:
        batch = hot_path.TrainPathBatch(frames, maps, top_diff_seed=3)
        np.random.seed(11)
        batch.setup()
    finally:
        for k, v in saved.items():
            cfg.TRAIN[k] = v
    first = batch.snapshot()
    for t in (batch.rpn_labels, batch.rois_3d, batch.bbox_targets, batch.tops["bev"][0], batch.bottom_diff["rgb"], batch.rois["fv"],
              batch.rois["rgb"]):
        t.fill_(7.0)                                            # the replay must rewrite everything
    batch.run()
    batch.run()
    torch.cuda.synchronize()
    again = batch.snapshot()
    for a, b in zip(first, again):
        assert np.array_equal(a, b, equal_nan=True)
    # ---- oracle, frame by frame, same seed and draw order
    np.random.seed(11)
    off = 0
    maps_h = {v: m.cpu().numpy() for v, m in maps.items()}
    exp = {v: [] for v in hot_path.VIEWS}
    for b in range(B):
        prob, pred, info, calib, (gt_bv, gt_3d, gt_cnr) = frames[b]
        bv, img, b3 = oracle.proposal_layer_3d(prob, pred, info, calib, "TRAIN", [8, ], cfg={"TRAIN": TRAIN_CFG})
        n = bv.shape[0]
        assert batch.num_proposals[b] == n
        lab, tg, anc, anc3 = oracle.anchor_target_layer(np.zeros((1, 76, 76, 8), np.float32), gt_bv, gt_3d, info, [8, ])
        assert np.array_equal(batch.rpn_labels[b].cpu().numpy(), lab) and np.array_equal(batch.rpn_targets[b].cpu().numpy(), tg)
        m = int(batch.n_anchors[b].item())
        assert m == anc.shape[0] and np.array_equal(batch.anchors[b, :m].cpu().numpy(), anc)
        assert np.array_equal(batch.anchors_3d[b, :m].cpu().numpy(), anc3)
        r_bv, r_img, r_lab, r_tg, r_3d = oracle.proposal_target_layer_3d(bv, b3, gt_bv, gt_3d, gt_cnr, calib, 2, train=o_train)
        S = r_bv.shape[0]
        assert batch.S[b] == S and 0 < S <= 128 and (S == 128 or not yml)
        r_bv[:, 0] = b; r_img[:, 0] = b; r_3d[:, 0] = b           # batched extension: the frame index
        sl = slice(off, off + S)
        assert np.array_equal(batch.rois["bev"][sl].cpu().numpy(), r_bv) and np.array_equal(batch.rois["rgb"][sl].cpu().numpy(), r_img)
        assert np.array_equal(batch.labels[sl].cpu().numpy(), r_lab) and np.array_equal(batch.bbox_targets[sl].cpu().numpy(), r_tg)
        assert np.array_equal(batch.rois_3d[sl].cpu().numpy(), r_3d)
        exp["bev"].append(r_bv); exp["rgb"].append(r_img); exp["fv"].append(oracle.rois_3d_to_fv(r_3d))
        off += S
    assert off == batch.num_rois
    # (the RoiPool pair keeps its argmax plane as private 16-bit codes: decoded for the comparison)
    dec = ops.roi_pool_argmax_decode([(batch.maps[v], batch.rois[v], 0.125) for v in hot_path.VIEWS], [batch.tops[v] for v in hot_path.VIEWS], 7, 7)
    for v, am in zip(hot_path.VIEWS, dec):
        rois = np.concatenate(exp[v])
        assert np.array_equal(batch.rois[v].cpu().numpy(), rois), v
        o_top, o_am = oracle.roi_pool(maps_h[v], rois, 7, 7, 0.125)
        assert np.array_equal(batch.tops[v][0].cpu().numpy(), o_top) and np.array_equal(am.cpu().numpy(), o_am), v
        want = oracle.roi_pool_grad(maps_h[v], rois, o_am, batch.top_diff[v].cpu().numpy(), 7, 7, 0.125)
        assert np.array_equal(batch.bottom_diff[v].cpu().numpy(), want), v
    # the TEST-cfg batch (configs[1] / configs[4] path) binds and replays too
    tb = hot_path.TestPathBatch([f[:4] for f in frames], maps).setup()
    tb.run()
    torch.cuda.synchronize()
    for b in range(B):
        bv, img, b3 = oracle.proposal_layer_3d(*frames[b][:4], "TEST", [8, ], cfg={"TEST": TEST_CFG})
        n = bv.shape[0]
        assert int(tb.prop[3][b].item()) == n and np.array_equal(tb.prop[0][b, :n, 1:].cpu().numpy(), bv[:, 1:])
    fv_want = oracle.rois_3d_to_fv(tb.prop[2].view(-1, 7).cpu().numpy())
    assert np.array_equal(tb.rois["fv"].cpu().numpy(), fv_want)
    o_top, o_am = oracle.roi_pool(maps_h["fv"], fv_want, 7, 7, 0.125)
    assert np.array_equal(tb.tops["fv"][0].cpu().numpy(), o_top) and np.array_equal(tb.tops["fv"][1].cpu().numpy(), o_am)


def test_three_view_graph_forward_backward(gpu):
    """row X1: the front view in the runnable graph (`MV3D_train_3view`): third trunk, rois_fv from the 3D ROIs, third
    RoiPool + tower; losses back-propagate into all three conv5_3 maps through RoiPoolGrad."""
    torch, ops = gpu
    from mv3d_tf_amd.fast_rcnn.train_mv import total_loss
    from mv3d_tf_amd.networks import get_network
    net = get_network("MV3D_train_3view")
    assert net.views == 3 and "conv5_3_3" in net.params and net.params["cls_score"][0].shape == (2, 6144)
    rng = np.random.RandomState(1)
    _, _, info, calib, (gt_bv, gt_3d, gt_cnr) = synth.rpn_head(55, 76, 76, "peaky", return_gt=True)
    np.random.seed(3)
    with torch.no_grad():
        net.params["rpn_cls_score"][0].mul_(40.0)
    L = net.forward({"lidar_bv_data": (rng.random_sample((1, 608, 608, 9)) < 0.02).astype(np.float32),
                     "image_data": rng.randint(0, 255, (1, 96, 320, 3)).astype(np.float32),
                     "lidar_fv_data": rng.random_sample((1, 64, 512, 3)).astype(np.float32),
                     "im_info": info, "calib": calib, "gt_boxes_bv": gt_bv, "gt_boxes_3d": gt_3d, "gt_boxes_corners": gt_cnr})
    S = L["roi_data_3d"][0].shape[0]
    assert L["conv5_3_3"].shape == (1, 8, 64, 512) and L["roi_data_fv"].shape == (S, 5) and L["pool_5_3"].shape == (S, 7, 7, 512)
    fv = L["roi_data_fv"].cpu().numpy()
    assert fv[:, 1:].min() >= 0 and fv[:, [1, 3]].max() <= 511 and fv[:, [2, 4]].max() <= 63
    loss, _ = total_loss(L)
    loss.backward()
    for k in ("conv5_3", "conv5_3_3", "fc6_3"):        # (the 96 x 320 test image is smaller than the 375 x 1242 the image ROIs assume)
        g = net.params[k][0].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0, k
    t = get_network("MV3D_test_3view")
    Lt = t.forward({"lidar_bv_data": np.zeros((1, 608, 608, 9), np.float32), "image_data": np.zeros((1, 96, 320, 3), np.float32),
                    "lidar_fv_data": np.zeros((1, 64, 512, 3), np.float32), "im_info": info, "calib": calib})
    assert Lt["cls_prob"].shape[1] == 2 and Lt["pool_5_3"].shape[0] == Lt["cls_prob"].shape[0]


def test_config4_serving_step_as_one_hipgraph(gpu):
    """BASELINE configs[4] as it is stated: the WHOLE serving step -- trunks (f16 MFMA), RPN heads, proposal_layer_3d TEST cfg 6000 -> 300,
    FV ROIs, RoiPool x3, fusion head, box tail (lib/fast_rcnn/test_mv.py:149-264 for a batch) -- captured ONCE (ServeGraph), no host
    sync inside.  The replay equals the same fixed-shape step run eagerly, bit for bit, also after new inputs were written into the
    graph's buffers; the rows of a frame below its num_rois carry the hot path's exact ROIs of the ordinary (host-synchronised,
    variable-shape) forward and its scores within 16-bit GEMM tolerance (a 1200-row and an R-row GEMM may tile differently)."""
    torch, ops = gpu
    from mv3d_tf_amd.fast_rcnn import test_mv
    from mv3d_tf_amd.fast_rcnn.config import cfg
    from mv3d_tf_amd.networks import get_network
    B = 4
    saved = (cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N)
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 6000, 300
    try:
        net = get_network("MV3D_test_3view")
        net.amp_dtype, net.mfma_trunk = torch.float16, True
        with torch.no_grad():
            net.params["rpn_cls_score"][0].mul_(30.0)                # (spread scores: NMS keeps a frame-dependent number of boxes)

        def feed(seed):
            rng = np.random.RandomState(seed)
            return {"lidar_bv_data": torch.as_tensor(((rng.random_sample((B, 608, 608, 9)) < 0.03) * rng.uniform(0, 2.4, (B, 608, 608, 9))).astype(np.float32)).cuda(),
                    "image_data": torch.as_tensor((rng.randint(0, 255, (B, 375, 1242, 3)) - cfg.PIXEL_MEANS).astype(np.float32)).cuda(),
                    "lidar_fv_data": torch.as_tensor(rng.uniform(0, 1, (B, 64, 512, 3)).astype(np.float32)).cuda(),
                    "im_info": np.array([[608, 608, 1]] * B, np.float32), "calib": np.stack([synth.KITTI_CALIB] * B), "keep_prob": 1.0}

        f1, f2 = feed(1), feed(2)
        sg = test_mv.ServeGraph(net, f1)
        assert net.fixed_rois is False                               # (the flag is the graph's business only)
        keys = ("cls_prob", "bbox_pred", "rois_bv", "rois_img", "rois_3d", "corners", "pred_corners_r", "pred_bv", "num_rois")
        for f in (f1, f2, f1):
            got = sg.replay(f)
            sg.stream.synchronize()
            got = {k: got[k].clone() for k in keys}
            net.fixed_rois = True
            try:
                with torch.no_grad():
                    L = net.forward(f)
                    cnr, pr, pbv, _ = ops.box_detect_tail(L["rois"][2].contiguous(), L["bbox_pred"].contiguous(), 2)
            finally:
                net.fixed_rois = False
            want = {"cls_prob": L["cls_prob"], "bbox_pred": L["bbox_pred"], "rois_bv": L["rois"][0], "rois_img": L["rois"][1], "rois_3d": L["rois"][2],
                    "corners": cnr, "pred_corners_r": pr, "pred_bv": pbv, "num_rois": L["num_rois"]}
            for k in keys:
                assert torch.equal(got[k], want[k]), k
            cap = 300
            assert got["cls_prob"].shape == (B * cap, 2) and got["rois_3d"].shape == (B * cap, 7)
            num = got["num_rois"].cpu().numpy()
            assert (num > 0).all() and (num <= cap).all()
            # the ordinary forward (counts to the host, variable shapes): the same ROIs in the rows a frame owns, zero rows behind them
            with torch.no_grad():
                V = net.forward(f)
            off = 0
            for b in range(B):
                n = int(num[b])
                for k, j in (("rois_bv", 0), ("rois_img", 1), ("rois_3d", 2)):
                    rows = got[k][b * cap:b * cap + n]
                    assert torch.equal(rows, V["rois"][j][off:off + n]), (k, b)
                    assert not got[k][b * cap + n:(b + 1) * cap].any()
                assert torch.allclose(got["cls_prob"][b * cap:b * cap + n], V["cls_prob"][off:off + n], atol=5e-3)
                off += n
            assert off == V["rois"][0].shape[0]
        dets = sg.detections()
        assert len(dets) == B and all(d[0].shape[0] == int(n) and d[1].shape == (int(n), 8) for d, n in zip(dets, sg.out["num_rois"].cpu().numpy()))
    finally:
        cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = saved
