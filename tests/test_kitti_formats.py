"""KITTI calib / label files -> ground-truth encodings and gt blobs, bit for bit (values and dtypes)
against what the reference's own loader (lib/datasets/kitti_mv3d.py) and get_minibatch
(lib/roi_data_layer/minibatch_mv3d.py) produced for the same files (tests/golden/kitti_label.npz; the file contents are
stored in the fixture as text).  CPU tests: the oracle's restatement of the geometry and the host-side parsing; `gpu` tests:
the product (text table -> mv3d_gt_encode on the device -> roidb entry)."""
import os

import numpy as np
import pytest

from conftest import golden
from mv3d_tf_amd.datasets import gt_blobs, load_kitti_calib, pack_calib

ANN_KEYS = ("ry", "lwh", "boxes", "boxes_bv", "boxes_3D_cam", "boxes_3D", "boxes3D_cam_corners", "boxes_corners",
            "gt_classes", "gt_overlaps", "xyz", "alphas")


def _tree(tmp_path, g):
    root = tmp_path / "KITTI"
    for sub in ("ImageSets", "object/training/calib", "object/training/label_2", "object/training/image_2",
                "object/training/lidar_bv"):
        os.makedirs(root / sub)
    n = int(g["n_frames"])
    for i in range(n):
        idx = "%06d" % i
        (root / "object/training/label_2" / (idx + ".txt")).write_text(str(g["labels_txt_%d" % i]))
        (root / "object/training/calib" / (idx + ".txt")).write_text(str(g["calib_txt_%d" % i]))
        (root / "object/training/image_2" / (idx + ".png")).write_bytes(b"")
        np.save(root / "object/training/lidar_bv" / (idx + ".npy"), np.zeros((8, 9, 9), np.float32))
    (root / "ImageSets" / "train.txt").write_text("".join("%06d\n" % i for i in range(n)))
    return str(root), n


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def _cars(g, i):
    lines = [ln for ln in str(g["labels_txt_%d" % i]).splitlines() if ln.split(" ")[0] == "Car"]
    return lines, [float(ln.split(" ")[14]) for ln in lines]


def test_oracle_gt_encode_matches_reference(oracle):
    g = golden("kitti_label")
    for i in range(int(g["n_frames"])):
        tr = g["calib_%d" % i][3].reshape(3, 4).astype(np.float32)
        _, ry = _cars(g, i)
        cam, lid, b3, bv = oracle.gt_encode(g["ann%d_boxes_3D_cam" % i], ry, tr)
        assert _same(cam, g["ann%d_boxes3D_cam_corners" % i]) and _same(lid, g["ann%d_boxes_corners" % i])
        assert _same(b3, g["ann%d_boxes_3D" % i]) and _same(bv, g["ann%d_boxes_bv" % i])


def test_calib_parsing_and_gt_blobs_host_side(tmp_path):
    g = golden("kitti_label")
    p = tmp_path / "c.txt"
    p.write_text(str(g["calib_txt_0"]))
    c = load_kitti_calib(str(p))
    assert c["P2"].dtype == np.float32 and c["Tr_velo2cam"].shape == (3, 4) and c["R0"].shape == (3, 3)
    assert _same(pack_calib(c), g["calib_0"])                          # (4,12) f64 table of f32 values
    for i in range(int(g["n_frames"])):
        ann = {k: g["ann%d_%s" % (i, k)] for k in ANN_KEYS}
        blobs = gt_blobs(ann, (8, 9, 9))
        for k in ("gt_boxes", "gt_boxes_bv", "gt_boxes_3d", "gt_boxes_corners", "im_info"):
            assert _same(blobs[k], g["blob%d_%s" % (i, k)]), (i, k)
    # BEV boxes are integral pixel coordinates, x1 <= x2 (the (+,+) corner comes first)
    bv = g["ann0_boxes_bv"]
    assert np.array_equal(bv, np.round(bv)) and (bv[:, 0] <= bv[:, 2]).all() and (bv[:, 1] <= bv[:, 3]).all()


@pytest.mark.gpu
def test_kitti_label_and_calib_match_reference(tmp_path, oracle):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from mv3d_tf_amd import build
    build.build()
    from mv3d_tf_amd.datasets import kitti_mv3d, parse_kitti_labels
    g = golden("kitti_label")
    root, n = _tree(tmp_path, g)
    db = kitti_mv3d("train", root)
    assert db.num_classes == 2 and db.image_index == ["%06d" % i for i in range(n)] and db.num_images == n
    roidb = db.gt_roidb()
    for i in range(n):
        cal = db.calib_at(i)
        assert _same(cal, g["calib_%d" % i]) and _same(cal, g["blob%d_calib" % i])
        ann = roidb[i]
        for k in ANN_KEYS:
            got = ann[k].toarray() if k == "gt_overlaps" else ann[k]
            assert _same(got, g["ann%d_%s" % (i, k)]), (i, k)
        assert ann["flipped"] is False
        blobs = gt_blobs(ann, (8, 9, 9))
        for k in ("gt_boxes", "gt_boxes_bv", "gt_boxes_3d", "gt_boxes_corners", "im_info"):
            assert _same(blobs[k], g["blob%d_%s" % (i, k)]), (i, k)
        assert db.image_path_at(i).endswith("image_2/%06d.png" % i) and db.lidar_path_at(i).endswith("lidar_bv/%06d.npy" % i)
    # frame 2 has no object of a known class: empty encodings, still well-formed
    assert roidb[2]["boxes"].shape == (0, 4) and roidb[2]["gt_overlaps"].shape == (0, 2)
    assert gt_blobs(roidb[2], (8, 9, 9))["gt_boxes_3d"].shape == (0, 7)
    # the device kernel against the oracle on many random objects (incl. yaw at +-pi, tiny and huge boxes)
    rng = np.random.RandomState(4)
    G = 3000
    box = np.stack([rng.uniform(-40, 40, G), rng.uniform(0.5, 2.5, G), rng.uniform(0.5, 80, G), rng.uniform(0.3, 12, G),
                    rng.uniform(0.3, 3, G), rng.uniform(0.5, 4, G)], 1).astype(np.float32)
    ry = rng.uniform(-np.pi, np.pi, G); ry[:4] = [np.pi, -np.pi, 0.0, np.pi / 2]
    tr = g["calib_0"][3].reshape(3, 4).astype(np.float32)
    lines = ["Car 0 0 0 1 2 3 4 %r %r %r %r %r %r %r" % (float(b[5]), float(b[4]), float(b[3]), float(b[0]), float(b[1]), float(b[2]), float(r))
             for b, r in zip(box, ry)]
    ann = parse_kitti_labels(lines, tr, {"__background__": 0, "Car": 1}, 2)
    cam, lid, b3, bv = oracle.gt_encode(box, ry, tr)
    assert _same(ann["boxes3D_cam_corners"], cam) and _same(ann["boxes_corners"], lid)
    assert _same(ann["boxes_3D"], b3) and _same(ann["boxes_bv"], bv)


@pytest.mark.gpu
def test_get_training_roidb_to_train_net_on_a_kitti_tree(tmp_path, capsys):
    """the documented entry chain with the DEFAULT config (USE_FLIPPED defaulted to True and the chain raised):
    kitti_mv3d(tree) -> get_training_roidb -> filter_roidb -> train_net, two iterations on the fixture's label / calib files
    with synthetic BEV maps and images."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from mv3d_tf_amd import build
    build.build()
    from mv3d_tf_amd.datasets import kitti_mv3d
    from mv3d_tf_amd.fast_rcnn import train_mv
    from mv3d_tf_amd.fast_rcnn.config import cfg
    from mv3d_tf_amd.networks import get_network
    assert cfg.TRAIN.USE_FLIPPED is False and cfg.TRAIN.MAX_SIZE == 2000          # lib/fast_rcnn/config.py:84,64
    g = golden("kitti_label")
    root, n = _tree(tmp_path, g)
    rng = np.random.RandomState(0)
    for i in range(n):                                                            # frames the graph can run on
        np.save(os.path.join(root, "object/training/lidar_bv/%06d.npy" % i), (rng.random_sample((608, 608, 9)) < 0.02).astype(np.float32))
        np.save(os.path.join(root, "object/training/image_2/%06d.npy" % i), rng.randint(0, 255, (96, 320, 3)).astype(np.uint8))
    db = kitti_mv3d("train", root)
    roidb = train_mv.get_training_roidb(db)
    assert len(roidb) == n and all("max_overlaps" in e and "calib" in e for e in roidb)
    for i, e in enumerate(roidb):                                                 # (.png placeholders of the fixture tree -> the arrays)
        e["image_path"] = os.path.join(root, "object/training/image_2/%06d.npy" % i)
    kept = train_mv.filter_roidb(roidb)
    assert 0 < len(kept) < n                                                      # the frame without a known class is dropped
    saved = (cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.DISPLAY)
    cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.DISPLAY = 1, 1
    try:
        np.random.seed(cfg.RNG_SEED)
        hist = train_mv.train_net(get_network("MV3D_train"), db, roidb, str(tmp_path / "out"), max_iters=2)
    finally:
        cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.DISPLAY = saved
    assert len(hist) == 2 and all(np.isfinite(h[0]) for h in hist)
    assert "iter: 2 / 2, total loss: " in capsys.readouterr().out
