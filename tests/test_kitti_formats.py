"""SURVEY §8(f) rank 3 (host logic, no GPU): KITTI calib / label files -> ground-truth encodings and gt blobs,
bit-for-bit against what the reference's own loader (lib/datasets/kitti_mv3d.py) and get_minibatch
(lib/roi_data_layer/minibatch_mv3d.py) produced for the same files (tests/golden/kitti_label.npz; the file
contents are stored in the fixture as text)."""
import os

import numpy as np

from conftest import golden
from mv3d_tf_amd.datasets import gt_blobs, kitti_mv3d, load_kitti_calib, pack_calib, parse_kitti_labels

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


def test_kitti_label_and_calib_match_reference(tmp_path):
    g = golden("kitti_label")
    root, n = _tree(tmp_path, g)
    db = kitti_mv3d("train", root)
    assert db.num_classes == 2 and db.image_index == ["%06d" % i for i in range(n)]
    roidb = db.gt_roidb()
    for i in range(n):
        cal = db.calib_at(i)
        assert _same(cal, g["calib_%d" % i])                              # (4,12) f64 table of f32 values
        ann = roidb[i]
        for k in ANN_KEYS:
            got = ann[k].toarray() if k == "gt_overlaps" else ann[k]
            assert _same(got, g["ann%d_%s" % (i, k)]), (i, k)
        assert ann["flipped"] is False
        blobs = gt_blobs(ann, (8, 9, 9))
        for k in ("gt_boxes", "gt_boxes_bv", "gt_boxes_3d", "gt_boxes_corners", "im_info"):
            assert _same(blobs[k], g["blob%d_%s" % (i, k)]), (i, k)
        assert _same(cal, g["blob%d_calib" % i])
        assert db.image_path_at(i).endswith("image_2/%06d.png" % i) and db.lidar_path_at(i).endswith("lidar_bv/%06d.npy" % i)
    # frame 2 has no object of a known class: empty encodings, still well-formed
    assert roidb[2]["boxes"].shape == (0, 4) and roidb[2]["gt_overlaps"].shape == (0, 2)
    assert gt_blobs(roidb[2], (8, 9, 9))["gt_boxes_3d"].shape == (0, 7)


def test_kitti_pieces_standalone(tmp_path):
    g = golden("kitti_label")
    p = tmp_path / "c.txt"
    p.write_text(str(g["calib_txt_0"]))
    c = load_kitti_calib(str(p))
    assert c["P2"].dtype == np.float32 and c["Tr_velo2cam"].shape == (3, 4) and c["R0"].shape == (3, 3)
    assert _same(pack_calib(c), g["calib_0"])
    ann = parse_kitti_labels(str(g["labels_txt_0"]).splitlines(True), c["Tr_velo2cam"], {"__background__": 0, "Car": 1}, 2)
    assert _same(ann["boxes_bv"], g["ann0_boxes_bv"]) and _same(ann["boxes_corners"], g["ann0_boxes_corners"])
    # BEV boxes are integral pixel coordinates inside (or just around) the 601-pixel map, x1 < x2 (+,+ corner first)
    bv = ann["boxes_bv"]
    assert np.array_equal(bv, np.round(bv)) and (bv[:, 0] <= bv[:, 2]).all() and (bv[:, 1] <= bv[:, 3]).all()
