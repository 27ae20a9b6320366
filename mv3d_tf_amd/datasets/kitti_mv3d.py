"""KITTI object files -> MV3D ground truth (lib/datasets/kitti_mv3d.py:63-75, 151-306) and the gt part of the
training blobs (lib/roi_data_layer/minibatch_mv3d.py:47-75).

  <kitti>/object/{training,testing}/calib/000000.txt     P0..P3, R0_rect, Tr_velo_to_cam, Tr_imu_to_velo
  <kitti>/object/training/label_2/000000.txt             type trunc occ alpha x1 y1 x2 y2 h w l tx ty tz ry
  <kitti>/object/training/{image_2/*.png, lidar_bv/*.npy}, <kitti>/ImageSets/<set>.txt

`kitti_mv3d` keeps the reference class's accessors (image_index, num_classes, image_path_at, lidar_path_at, calib_at,
gt_roidb); evaluation, caching and the proposal-recall statistics of the reference are out of scope."""
import os

import numpy as np
import scipy.sparse

from ..utils.transform import camera_to_lidar_cnr, computeCorners3D, lidar_3d_to_bv, lidar_cnr_to_3d


def load_kitti_calib(path):
    """lib/datasets/kitti_mv3d.py:151-192: rows 2..5 of the calib file (P2, P3, R0_rect, Tr_velo_to_cam), f32."""
    with open(path) as f:
        rows = f.readlines()
    vals = [np.array(rows[k].strip().split(' ')[1:], dtype=np.float32) for k in (2, 3, 4, 5)]
    return {'P2': vals[0].reshape(3, 4), 'P3': vals[1].reshape(3, 4), 'R0': vals[2].reshape(3, 3),
            'Tr_velo2cam': vals[3].reshape(3, 4)}


def pack_calib(c):
    """lib/datasets/kitti_mv3d.py:63-75 (calib_at): the (4, 12) table the layers take -- rows P2, P3, R0 (9 values,
    zero padded), Tr_velo_to_cam."""
    out = np.zeros((4, 12))
    out[0, :] = c['P2'].reshape(12)
    out[1, :] = c['P3'].reshape(12)
    out[2, :9] = c['R0'].reshape(9)
    out[3, :] = c['Tr_velo2cam'].reshape(12)
    return out


def parse_kitti_labels(lines, Tr, class_to_ind, num_classes):
    """lib/datasets/kitti_mv3d.py:194-306 for the label lines of one frame: objects whose type is not in
    class_to_ind are skipped; for the others the 2D box, the camera box, its 8 corners (camera and LIDAR frame), the
    LIDAR box and the BEV pixel box, all f32 like the reference's arrays."""
    n = len(lines)
    trans = np.zeros((n, 3), dtype=np.float32)
    rys = np.zeros((n), dtype=np.float32)
    lwh = np.zeros((n, 3), dtype=np.float32)
    boxes = np.zeros((n, 4), dtype=np.float32)
    boxes_bv = np.zeros((n, 4), dtype=np.float32)
    box_cam = np.zeros((n, 6), dtype=np.float32)
    box_lidar = np.zeros((n, 6), dtype=np.float32)
    cnr_cam = np.zeros((n, 24), dtype=np.float32)
    cnr_lidar = np.zeros((n, 24), dtype=np.float32)
    alphas = np.zeros((n), dtype=np.float32)
    classes = np.zeros((n), dtype=np.int32)
    overlaps = np.zeros((n, num_classes), dtype=np.float32)
    k = 0
    for line in lines:
        tok = line.strip().split(' ')
        cls = class_to_ind.get(tok[0].strip())
        if cls is None:
            continue
        alpha, x1, y1, x2, y2, h, w, l, tx, ty, tz, ry = (float(v) for v in tok[3:15])
        rys[k] = ry
        lwh[k, :] = [l, w, h]
        alphas[k] = alpha
        trans[k, :] = [tx, ty, tz]
        boxes[k, :] = [x1, y1, x2, y2]
        box_cam[k, :] = [tx, ty, tz, l, w, h]
        cam = computeCorners3D(box_cam[k, :], ry)                     # from the f32 row, yaw as Python float
        cnr_cam[k, :] = cam.reshape(24)
        cnr_lidar[k, :] = camera_to_lidar_cnr(cam, Tr)
        box_lidar[k, :] = lidar_cnr_to_3d(cnr_lidar[k, :], lwh[k, :])
        boxes_bv[k, :] = lidar_3d_to_bv(box_lidar[k, :])
        classes[k] = cls
        overlaps[k, cls] = 1.0
        k += 1
    return {'ry': rys[:k].copy(), 'lwh': lwh[:k].copy(), 'boxes': boxes[:k].copy(), 'boxes_bv': boxes_bv[:k].copy(),
            'boxes_3D_cam': box_cam[:k].copy(), 'boxes_3D': box_lidar[:k].copy(),
            'boxes3D_cam_corners': cnr_cam[:k].copy(), 'boxes_corners': cnr_lidar[:k].copy(),
            'gt_classes': classes[:k].copy(), 'gt_overlaps': scipy.sparse.csr_matrix(overlaps[:k]),
            'xyz': trans[:k].copy(), 'alphas': alphas[:k].copy(), 'flipped': False}


def gt_blobs(entry, lidar_bv_shape):
    """lib/roi_data_layer/minibatch_mv3d.py:47-75: the ground-truth blobs of one roidb entry -- (G,5) image boxes,
    (G,5) BEV boxes, (G,7) LIDAR boxes, (G,25) LIDAR corners, each with the class in the last column, and im_info =
    (BEV height, BEV width, 1)."""
    sel = np.where(entry['gt_classes'] != 0)[0]
    cls = entry['gt_classes'][sel]

    def with_class(a, width):
        out = np.empty((len(sel), width + 1), dtype=np.float32)
        out[:, 0:width] = a[sel, :]
        out[:, width] = cls
        return out

    return {'gt_boxes': with_class(entry['boxes'] * 1, 4), 'gt_boxes_bv': with_class(entry['boxes_bv'], 4),
            'gt_boxes_3d': with_class(entry['boxes_3D'], 6), 'gt_boxes_corners': with_class(entry['boxes_corners'], 24),
            'im_info': np.array([[lidar_bv_shape[0], lidar_bv_shape[1], 1]], dtype=np.float32)}


class kitti_mv3d(object):
    """The accessors of lib/datasets/kitti_mv3d.py the train / test entry points use."""

    def __init__(self, image_set, kitti_path):
        self._image_set = image_set
        self._kitti_path = kitti_path
        self._data_path = os.path.join(kitti_path, 'object')
        self._classes = ('__background__', 'Car')
        self._class_to_ind = dict(zip(self._classes, range(len(self._classes))))
        self._image_ext, self._lidar_ext = '.png', '.npy'
        assert os.path.exists(self._kitti_path), 'KITTI path does not exist: {}'.format(self._kitti_path)
        assert os.path.exists(self._data_path), 'Path does not exist: {}'.format(self._data_path)
        set_file = os.path.join(kitti_path, 'ImageSets', image_set + '.txt')
        assert os.path.exists(set_file), 'Path does not exist: {}'.format(set_file)
        with open(set_file) as f:
            self._image_index = [x.rstrip('\n') for x in f.readlines()]
        self.name = 'kitti_mv3d_' + image_set

    classes = property(lambda self: self._classes)
    num_classes = property(lambda self: len(self._classes))
    image_index = property(lambda self: self._image_index)

    def _split(self):
        return 'testing' if self._image_set == 'test' else 'training'

    def image_path_at(self, i):
        p = os.path.join(self._data_path, self._split(), 'image_2', self._image_index[i] + self._image_ext)
        assert os.path.exists(p), 'Path does not exist: {}'.format(p)
        return p

    def lidar_path_at(self, i):
        p = os.path.join(self._data_path, self._split(), 'lidar_bv', self._image_index[i] + self._lidar_ext)
        assert os.path.exists(p), 'Path does not exist: {}'.format(p)
        return p

    def _load_kitti_calib(self, index):
        return load_kitti_calib(os.path.join(self._data_path, self._split(), 'calib', index + '.txt'))

    def calib_at(self, i):
        return pack_calib(self._load_kitti_calib(str(i).zfill(6)))          # (sic: by position, kitti_mv3d.py:67)

    def _load_kitti_annotation(self, index):
        with open(os.path.join(self._data_path, 'training/label_2', index + '.txt')) as f:
            lines = f.readlines()
        return parse_kitti_labels(lines, self._load_kitti_calib(index)['Tr_velo2cam'], self._class_to_ind, self.num_classes)

    def gt_roidb(self):
        return [self._load_kitti_annotation(index) for index in self._image_index]
