"""On-disk formats either side of the hot path: KITTI calibration / label files ->
the ground-truth encodings the training layers consume."""
from .kitti_mv3d import kitti_mv3d, load_kitti_calib, pack_calib, parse_kitti_labels, gt_blobs  # noqa: F401
