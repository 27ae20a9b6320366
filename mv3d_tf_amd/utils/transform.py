"""Host-side geometry of the ground-truth encodings: the four helpers of
lib/utils/transform.py that lib/datasets/kitti_mv3d.py:_load_kitti_annotation chains per labelled object --
camera box -> 8 camera corners -> 8 LIDAR corners -> LIDAR box -> BEV pixel box.  A frame has a handful of
objects, so this is plain numpy on the host like the reference; the expressions keep the reference's dtypes
(which decide the rounding: f32 label values, f64 rotation, f32 inverse of Tr, f64 floor-divide) and the same
np.dot shapes.  Pinned bit-for-bit by tests/golden/kitti_label.npz.
"""
import numpy as np

# lib/utils/transform.py:3-11 (the same Python expressions: 60 // 0.1 is 599.0, so Xn = Yn = 600)
TOP_X_MIN, TOP_X_MAX, TOP_Y_MIN, TOP_Y_MAX, RES = 0, 60, -30, 30, 0.1
XN = int((TOP_X_MAX - TOP_X_MIN) // RES) + 1
YN = int((TOP_Y_MAX - TOP_Y_MIN) // RES) + 1


def _bv_coord(x, y):
    """lib/utils/transform.py:13-20 -- LIDAR metres -> BEV pixel (column, row), floor-divide in the operands' dtype."""
    return YN - (y - TOP_Y_MIN) // RES, XN - (x - TOP_X_MIN) // RES


def computeCorners3D(Boxex3D, ry):
    """lib/utils/transform.py:441-465: camera box (x, y, z, l, w, h) + yaw -> (3, 8) corners in camera coordinates
    (y down: the box stands on y, its top is at y - h)."""
    c, s = np.cos(ry), np.sin(ry)
    rot = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]).reshape((3, 3))
    length, width, height = Boxex3D[3:6]
    cx, cy, cz = Boxex3D[0:3]
    hl, hw = length / 2, width / 2
    local = np.vstack((np.array([hl, hl, -hl, -hl, hl, hl, -hl, -hl]),
                       np.array([0, 0, 0, 0, -height, -height, -height, -height]),
                       np.array([hw, -hw, -hw, hw, hw, -hw, -hw, hw])))
    out = np.dot(rot, local)
    out[0, :] = out[0, :] + cx
    out[1, :] = out[1, :] + cy
    out[2, :] = out[2, :] + cz
    return out


def camera_to_lidar_cnr(pts_3D, P):
    """lib/utils/transform.py:502-524: camera corners (3, 8) [or (1, 24)] -> LIDAR corners (1, 24) with the inverse
    rotation of Tr_velo_to_cam and the reference's own translation column (-P[1,3], -P[2,3], P[0,3]); the corners
    get a homogeneous 0, so the translation is in fact dropped (kept: reference behaviour)."""
    if pts_3D.shape[1] == 24:
        pts_3D = pts_3D.reshape((3, 8))
    homo = np.vstack((pts_3D, np.zeros(8)))
    assert homo.shape == (4, 8)
    inv = np.linalg.inv(P[:, :3])
    shift = np.zeros((3, 1))
    shift[0] = -P[1, 3]
    shift[1] = -P[2, 3]
    shift[2] = P[0, 3]
    return np.dot(np.hstack((inv, shift)), homo)[:3, :].reshape(-1, 24)


def lidar_cnr_to_3d(corners, lwh):
    """lib/utils/transform.py:172-187: LIDAR corners -> (x, y, z, l, w, h): centre = mean of the 8 corners."""
    if corners.shape[0] == 24:
        box = np.zeros(6)
        box[:3] = corners.reshape((3, 8)).mean(1)
        box[3:] = lwh
        return box
    box = np.zeros((corners.shape[0], 6))
    box[:, :3] = corners.reshape((-1, 3, 8)).mean(2)
    box[:, 3:] = lwh
    return box


def lidar_3d_to_bv(rois_3d):
    """lib/utils/transform.py:113-142: LIDAR box -> BEV pixel box (x1, y1, x2, y2), f32.  (The device kernels carry
    the batched form of this for proposals; this is the host form the dataset loader calls per object.)"""
    if len(rois_3d.shape) == 1:
        out = np.zeros(4)
        out[0] = rois_3d[0] + rois_3d[3] * 0.5
        out[1] = rois_3d[1] + rois_3d[4] * 0.5
        out[2] = rois_3d[0] - rois_3d[3] * 0.5
        out[3] = rois_3d[1] - rois_3d[4] * 0.5
        out[0], out[1] = _bv_coord(out[0], out[1])
        out[2], out[3] = _bv_coord(out[2], out[3])
    else:
        out = np.zeros((rois_3d.shape[0], 4))
        out[:, 0] = rois_3d[:, 0] + rois_3d[:, 3] * 0.5
        out[:, 1] = rois_3d[:, 1] + rois_3d[:, 4] * 0.5
        out[:, 2] = rois_3d[:, 0] - rois_3d[:, 3] * 0.5
        out[:, 3] = rois_3d[:, 1] - rois_3d[:, 4] * 0.5
        out[:, 0], out[:, 1] = _bv_coord(out[:, 0], out[:, 1])
        out[:, 2], out[:, 3] = _bv_coord(out[:, 2], out[:, 3])
    return out.astype(np.float32)
