"""-m gpu tests of the fresh-input training path (mv3d_tf_amd.train_path.TrainPathStream) and of the train graph wired to it:

  * N consecutive FRESH batches through submit() / finish() (pipelined: batch i + 1 submitted before batch i is finished)
    equal the oracle run frame by frame with the same numpy seed, draw for draw;
  * a batch-2 step of the MV3D_train graph goes through the batched C entries (`*_batch`, `*_views_pair`: the RoiPool pair with its private
    compact argmax plane: a call counter on the ctypes handle) and its target layers / ROIs equal the per-frame oracle."""
import numpy as np
import pytest

from mv3d_tf_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from mv3d_tf_amd import build
    build.build()
    return torch


def _oracle_frame(oracle, frame, b, train):
    prob, pred, info, calib, (gt_bv, gt_3d, gt_cnr) = frame
    lab, tg, anc, anc3 = oracle.anchor_target_layer(np.zeros((1, prob.shape[1], prob.shape[2], 8), np.float32), gt_bv, gt_3d, info, [8, ])
    bv, img, b3 = oracle.proposal_layer_3d(prob, pred, info, calib, "TRAIN", [8, ])
    r_bv, r_img, r_lab, r_tg, r_3d = oracle.proposal_target_layer_3d(bv, b3, gt_bv, gt_3d, gt_cnr, calib, 2, train=train)
    for a in (r_bv, r_img, r_3d):
        a[:, 0] = b
    fv = oracle.rois_3d_to_fv(r_3d)
    return dict(labels=lab, targets=tg, anchors=anc, anchors_3d=anc3, n_prop=bv.shape[0], rois_bv=r_bv, rois_img=r_img, rois_lab=r_lab,
                rois_tg=r_tg, rois_3d=r_3d, rois_fv=fv)


@pytest.mark.parametrize("yml", [False, True])
def test_fresh_batches_pipelined_vs_oracle(gpu, oracle, yml):
    torch = gpu
    from mv3d_tf_amd.fast_rcnn.config import cfg
    from mv3d_tf_amd.train_path import TrainPathStream
    saved = {k: cfg.TRAIN[k] for k in ("BG_THRESH_LO", "BG_THRESH_HI", "FG_THRESH")}
    o_train = dict(oracle.TRAIN)
    if yml:                                                     # experiments/cfgs/faster_rcnn_end2end.yml:10-12
        o_train.update(BG_THRESH_LO=0.0, BG_THRESH_HI=0.5, FG_THRESH=0.7)
        cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.BG_THRESH_HI, cfg.TRAIN.FG_THRESH = 0.0, 0.5, 0.7
    try:
        B, NB = 2, 4
        dev = torch.device("cuda")
        # (yml case: every slot on its own stream -- the batches in flight overlap on the device as well)
        path = TrainPathStream(B, 76, 76, dev, depth=2, streams=[torch.cuda.Stream(), torch.cuda.Stream()] if yml else None)
        batches = [[synth.rpn_head(4100 + 10 * i + b, 76, 76, "peaky", return_gt=True) for b in range(B)] for i in range(NB)]
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)

        def upload(frames):
            return (t(np.concatenate([f[0] for f in frames])), t(np.concatenate([f[1] for f in frames])),
                    t(np.concatenate([f[2] for f in frames])), t(np.stack([f[3] for f in frames])),
                    [tuple(t(a) for a in f[4]) for f in frames])

        np.random.seed(21)
        got = []
        slot = path.submit(*upload(batches[0]))
        for i in range(NB):
            nxt = path.submit(*upload(batches[i + 1])) if i + 1 < NB else None      # stage 1 of batch i + 1 is in flight ...
            out = path.finish(slot)                                                 # ... while batch i draws and runs stage 2
            torch.cuda.synchronize()
            got.append({k: (v.cpu().numpy().copy() if isinstance(v, torch.Tensor) else
                            ({kk: vv.cpu().numpy().copy() for kk, vv in v.items()} if isinstance(v, dict) else v)) for k, v in out.items()})
            slot = nxt
        # oracle: same seed, frames in the same order, anchor draws before proposal draws within a frame
        np.random.seed(21)
        for i in range(NB):
            g = got[i]
            off = 0
            for b in range(B):
                w = _oracle_frame(oracle, batches[i][b], b, o_train)
                assert np.array_equal(g["rpn_labels"][b], w["labels"]) and np.array_equal(g["rpn_targets"][b], w["targets"])
                m = int(g["n_anchors"][b])
                assert m == w["anchors"].shape[0] and np.array_equal(g["anchors"][b, :m], w["anchors"])
                assert np.array_equal(g["anchors_3d"][b, :m], w["anchors_3d"])
                assert g["num_proposals"][b] == w["n_prop"]
                S = w["rois_bv"].shape[0]
                assert g["S"][b] == S
                sl = slice(off, off + S)
                assert np.array_equal(g["rois"]["bev"][sl], w["rois_bv"]) and np.array_equal(g["rois"]["rgb"][sl], w["rois_img"])
                assert np.array_equal(g["rois"]["fv"][sl], w["rois_fv"]) and np.array_equal(g["rois_3d"][sl], w["rois_3d"])
                assert np.array_equal(g["labels"][sl], w["rois_lab"]) and np.array_equal(g["bbox_targets"][sl], w["rois_tg"])
                off += S
            assert off == g["rois"]["bev"].shape[0]
    finally:
        for k, v in saved.items():
            cfg.TRAIN[k] = v


@pytest.mark.parametrize("helper", [True, False])
def test_host_stage_without_helper_thread_and_with_spilled_flags(gpu, oracle, helper, monkeypatch):
    """The C object's host stage (a) run by finish() itself (helper_thread = 0) and (b) with a report head too small for a frame's
    foreground flags (the flags are then fetched from the device before the third anchor draw): same outputs, draw for draw."""
    torch = gpu
    from mv3d_tf_amd import train_path
    monkeypatch.setattr(train_path, "_HEAD", 32 + 8)                    # 8 flags travel with the counts; the frames have more positives
    dev = torch.device("cuda")
    B = 2
    path = train_path.TrainPathStream(B, 76, 76, dev, depth=2, async_draws=helper)
    frames = [synth.rpn_head(5200 + b, 76, 76, "peaky", return_gt=True) for b in range(B)]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    args = (t(np.concatenate([f[0] for f in frames])), t(np.concatenate([f[1] for f in frames])), t(np.concatenate([f[2] for f in frames])),
            t(np.stack([f[3] for f in frames])), [tuple(t(a) for a in f[4]) for f in frames])
    np.random.seed(33)
    out = path.finish(path.submit(*args))
    torch.cuda.synchronize()
    state_after = np.random.get_state()[1].copy()
    np.random.seed(33)
    off = 0
    for b in range(B):
        w = _oracle_frame(oracle, frames[b], b, dict(oracle.TRAIN))
        assert np.array_equal(out["rpn_labels"][b].cpu().numpy(), w["labels"])
        m = int(out["n_anchors"][b])
        assert m == w["anchors"].shape[0] and np.array_equal(out["anchors"][b, :m].cpu().numpy(), w["anchors"])
        S = w["rois_bv"].shape[0]
        assert out["S"][b] == S and np.array_equal(out["rois"]["bev"][off:off + S].cpu().numpy(), w["rois_bv"])
        assert np.array_equal(out["bbox_targets"][off:off + S].cpu().numpy(), w["rois_tg"])
        off += S
    assert np.array_equal(np.random.get_state()[1], state_after)         # the generator ends where the reference's draws leave it
    # a slot can be submitted only once before it is finished; finish() of an idle slot is refused by the library
    s0 = path.submit(*args)
    s1 = path.submit(*args)
    with pytest.raises(RuntimeError):
        path.submit(*args)
    path.finish(s0), path.finish(s1)
    from mv3d_tf_amd._lib import Mv3dError
    with pytest.raises(Mv3dError):
        path.finish(s0)
    path.close()
    path.close()


def test_batch2_train_graph_uses_batched_entries_and_matches_oracle(gpu, oracle):
    torch = gpu
    from mv3d_tf_amd import _lib
    from mv3d_tf_amd.fast_rcnn.train_mv import total_loss
    from mv3d_tf_amd.networks import get_network
    L_ = _lib.lib()
    # (the target layers + proposal layer of a batch are ONE submit / finish pair of the C object mv3d_train_path, which issues the
    # batched entries itself; none of the per-frame or per-stage entries is called from Python any more)
    counted = ("mv3d_train_path_submit", "mv3d_train_path_finish", "mv3d_roi_pool_forward_views_pair", "mv3d_roi_pool_backward_views_pair",
               "mv3d_roi_pool_forward_views", "mv3d_roi_pool_backward_views", "mv3d_proposal_3d", "mv3d_anchor_target_stage1_batch", "mv3d_anchor_target_stage2_batch",
               "mv3d_proposal_target_stage1_batch_devn", "mv3d_proposal_target_stage2_batch_devn",
               "mv3d_roi_pool_forward", "mv3d_roi_pool_backward", "mv3d_anchor_target_stage1", "mv3d_proposal_target_stage1")
    calls, orig = {k: [] for k in counted}, {}
    for name in counted:
        orig[name] = getattr(L_, name)

        def wrap(*a, _n=name):
            calls[_n].append(a)
            return orig[_n](*a)
        setattr(L_, name, wrap)
    try:
        net = get_network("MV3D_train")
        with torch.no_grad():
            net.params["rpn_cls_score"][0].mul_(40.0)
        rng = np.random.RandomState(5)
        B = 2
        gt = []
        for b in range(B):
            r = np.random.RandomState(30 + b)
            gt.append(synth.gt_cars(r, 3 + b))
        feed = {"lidar_bv_data": (rng.random_sample((B, 608, 608, 9)) < 0.02).astype(np.float32),
                "image_data": rng.randint(0, 255, (B, 96, 320, 3)).astype(np.float32),
                "im_info": np.array([[608, 608, 1]] * B, np.float32), "calib": np.stack([synth.KITTI_CALIB] * B),
                "gt_boxes_bv": [g[0] for g in gt], "gt_boxes_3d": [g[1] for g in gt], "gt_boxes_corners": [g[2] for g in gt]}
        np.random.seed(9)
        L = net.forward(feed)
        loss, parts = total_loss(L)
        loss.backward()
        torch.cuda.synchronize()
        # ---- every hot-path layer of the step ran ONCE, through the batched / multi-view entries
        for name in counted[:4]:
            assert len(calls[name]) == 1, (name, len(calls[name]))
        for name in counted[4:]:
            assert len(calls[name]) == 0, (name, len(calls[name]))
        assert len(L["roi_rows"]) == B
        fw = calls["mv3d_roi_pool_forward_views_pair"][0]
        bw = calls["mv3d_roi_pool_backward_views_pair"][0]
        assert fw[0] == 2 and bw[0] == 2 and bw[4] is None and bw[5] == 0        # (the pair WITHOUT a workspace: RoiPoolGrad as one launch)
        # ---- the layers equal the per-frame oracle on the graph's own RPN head (same seed, draw for draw)
        prob = L["rpn_cls_prob_reshape"].detach().cpu().numpy()
        pred = L["rpn_bbox_pred"].detach().cpu().numpy()
        np.random.seed(9)
        labels, targets = L["rpn_data"][0].cpu().numpy(), L["rpn_data"][1].cpu().numpy()
        rois_bv, rois_img, rlab, rtg, r3d = (t.cpu().numpy() for t in L["roi_data_3d"])
        off = 0
        for b in range(B):
            frame = (prob[b:b + 1], pred[b:b + 1], feed["im_info"][b:b + 1], feed["calib"][b], gt[b])
            w = _oracle_frame(oracle, frame, b, dict(oracle.TRAIN))
            assert np.array_equal(labels[b], w["labels"]) and np.array_equal(targets[b], w["targets"])
            S = w["rois_bv"].shape[0]
            assert L["roi_rows"][b] == S
            sl = slice(off, off + S)
            assert np.array_equal(rois_bv[sl], w["rois_bv"]) and np.array_equal(rois_img[sl], w["rois_img"])
            assert np.array_equal(rlab[sl], w["rois_lab"]) and np.array_equal(rtg[sl], w["rois_tg"]) and np.array_equal(r3d[sl], w["rois_3d"])
            off += S
        assert L["pool_5"].shape == (off, 7, 7, 512) and L["cls_score"].shape == (off, 2)
        # RoiPool of the graph == oracle on the graph's own conv5_3 maps
        for name, conv, rois in (("pool_5", "conv5_3", rois_bv), ("pool_5_2", "conv5_3_2", rois_img)):
            fmap = np.ascontiguousarray(L[conv].detach().cpu().numpy())
            top, _ = oracle.roi_pool(fmap, rois, 7, 7, 0.125)
            assert np.array_equal(L[name].detach().cpu().numpy(), top), name
        g = net.params["conv5_3"][0].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
        assert all(np.isfinite(float(v.detach())) for v in parts)
    finally:
        for name in counted:
            setattr(L_, name, orig[name])


def test_dense_frame_grows_the_ground_truth_capacity(gpu, oracle):
    """a frame with more than 64 ground-truth boxes must not abort training -- the graph rebuilds its path slots
    with a larger capacity, and the path with 100 boxes equals the oracle."""
    torch = gpu
    from mv3d_tf_amd.networks import get_network
    from mv3d_tf_amd.train_path import TrainPathStream
    net = get_network("MV3D_train")
    assert net._train_path(1, 76, 76, 3).max_gt == 64
    assert net._train_path(1, 76, 76, 100).max_gt == 128
    assert net._train_path(1, 76, 76, 5).max_gt == 128                # (capacities only grow: no rebuild per sparse frame)
    dev = torch.device("cuda")
    prob, pred, info, calib, _ = synth.rpn_head(4321, 76, 76, "peaky", return_gt=True)
    gt = synth.gt_cars(np.random.RandomState(77), 100)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    path = TrainPathStream(1, 76, 76, dev, depth=1, max_gt=128)
    with pytest.raises(ValueError):
        TrainPathStream(1, 76, 76, dev, depth=1, max_gt=64).submit(t(prob), t(pred), t(info), t(calib[None]), [tuple(t(a) for a in gt)])
    np.random.seed(4)
    out = path.finish(path.submit(t(prob), t(pred), t(info), t(calib[None]), [tuple(t(a) for a in gt)]))
    torch.cuda.synchronize()
    np.random.seed(4)
    w = _oracle_frame(oracle, (prob, pred, info, calib, gt), 0, dict(oracle.TRAIN))
    assert np.array_equal(out["rpn_labels"][0].cpu().numpy(), w["labels"])
    assert np.array_equal(out["rpn_targets"][0].cpu().numpy(), w["targets"])
    assert np.array_equal(out["rois"]["bev"].cpu().numpy(), w["rois_bv"]) and np.array_equal(out["labels"].cpu().numpy(), w["rois_lab"])
    assert np.array_equal(out["bbox_targets"].cpu().numpy(), w["rois_tg"])
