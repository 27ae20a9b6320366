"""KITTI object files -> MV3D ground truth, and the gt blobs of a training frame.

  <kitti>/object/{training,testing}/calib/000000.txt     P0..P3, R0_rect, Tr_velo_to_cam, Tr_imu_to_velo
  <kitti>/object/training/label_2/000000.txt             type trunc occ alpha x1 y1 x2 y2 h w l tx ty tz ry
  <kitti>/object/training/{image_2/*.png, lidar_bv/*.npy}, <kitti>/ImageSets/<set>.txt

Interface kept from the reference (lib/datasets/kitti_mv3d.py): class `kitti_mv3d(image_set, kitti_path)` with
`image_index`, `num_classes`, `image_path_at`, `lidar_path_at`, `calib_at`, `gt_roidb`, and roidb entries with the same
keys / shapes / dtypes (:274-306); the (4,12) calibration table (:63-75); the gt blobs of
lib/roi_data_layer/minibatch_mv3d.py:47-75.

How it is done here: a frame's label file is parsed as ONE token table (no per-object Python loop); the numbers of the kept
objects go to the device in one upload, `mv3d_gt_encode` (csrc/gt_encode.hip) computes camera corners, LIDAR corners, LIDAR
box and BEV box for all objects at once, and one download brings the four arrays back.  Pinned bit for bit (values and
dtypes) by tests/golden/kitti_label.npz, which the reference's own loader produced.  Evaluation, caching and the
proposal-recall statistics of the reference class are out of scope."""
import os

import numpy as np
import scipy.sparse
import torch

from .. import ops
from ..fast_rcnn.config import cfg

# calibration file: the reference reads lines 2..5 by position (:158-168); KITTI writes them in this order
_CALIB_ROWS = ((2, 'P2', (3, 4)), (3, 'P3', (3, 4)), (4, 'R0', (3, 3)), (5, 'Tr_velo2cam', (3, 4)))


def load_kitti_calib(path):
    """{'P2' (3,4), 'P3' (3,4), 'R0' (3,3), 'Tr_velo2cam' (3,4)}, f32, from a KITTI calibration file."""
    with open(path) as f:
        text = f.read().splitlines()
    return {key: np.array(text[row].split()[1:], dtype=np.float32).reshape(shape) for row, key, shape in _CALIB_ROWS}


def pack_calib(c):
    """The (4, 12) table the layers take as `calib`: P2 | P3 | R0 (9 numbers + 3 zeros) | Tr_velo_to_cam, f64 holding the
    f32 file values (what `calib_at` returns in the reference)."""
    table = np.zeros((4, 12))
    for row, key in enumerate(('P2', 'P3', 'R0', 'Tr_velo2cam')):
        flat = c[key].ravel()
        table[row, :flat.size] = flat
    return table


def _device():
    return torch.device("cuda", cfg.GPU_ID)


def parse_kitti_labels(lines, Tr, class_to_ind, num_classes):
    """Annotation dict of one frame (roidb entry) from its label lines.  Objects whose type is not a key of `class_to_ind`
    are dropped.  Geometry on the device: `mv3d_gt_encode`."""
    table = [ln.split() for ln in lines if ln.strip()]
    table = [t for t in table if t[0] in class_to_ind]
    G = len(table)
    cls = np.array([class_to_ind[t[0]] for t in table], dtype=np.int32).reshape(G)
    # label columns 3..14 (KITTI devkit order: alpha | x1 y1 x2 y2 | h w l | tx ty tz | rotation_y) as the text's floats
    num = np.array([t[3:15] for t in table], dtype=np.float64).reshape(G, 12)
    alpha, box2d, hwl, xyz, ry = num[:, 0], num[:, 1:5], num[:, 5:8], num[:, 8:11], num[:, 11]
    lwh = hwl[:, ::-1]
    box_cam = np.ascontiguousarray(np.hstack([xyz, lwh]), dtype=np.float32)          # (tx, ty, tz, l, w, h) f32
    ann = {'ry': ry.astype(np.float32), 'lwh': lwh.astype(np.float32), 'boxes': box2d.astype(np.float32),
           'boxes_3D_cam': box_cam, 'gt_classes': cls, 'xyz': xyz.astype(np.float32), 'alphas': alpha.astype(np.float32),
           'flipped': False}
    onehot = np.zeros((G, num_classes), dtype=np.float32)
    onehot[np.arange(G), cls] = 1.0
    ann['gt_overlaps'] = scipy.sparse.csr_matrix(onehot)
    tr = np.ascontiguousarray(Tr, dtype=np.float32).reshape(3, 4)
    if G:
        dev = _device()
        cos_sin = np.stack([np.cos(ry), np.sin(ry)], axis=1)                          # f64, from the text's floats
        inv_rot = np.linalg.inv(tr[:, :3])                                            # f32 (LAPACK), once per frame
        d_box, d_inv, d_tr = ops.upload_packed([box_cam, inv_rot, tr], dev)
        d_cs = torch.from_numpy(np.ascontiguousarray(cos_sin)).to(dev)
        pack, spec, _ = ops.gt_encode(d_box, d_cs, d_inv.reshape(-1), d_tr.reshape(-1))
        cam, lid, box3d, bv = ops.unpack_host(pack, spec)
    else:
        cam, lid = np.zeros((0, 24), np.float32), np.zeros((0, 24), np.float32)
        box3d, bv = np.zeros((0, 6), np.float32), np.zeros((0, 4), np.float32)
    ann.update({'boxes3D_cam_corners': cam, 'boxes_corners': lid, 'boxes_3D': box3d, 'boxes_bv': bv})
    return ann


def gt_blobs(entry, lidar_bv_shape):
    """Ground-truth blobs of one roidb entry: image / BEV / LIDAR boxes and LIDAR corners of the foreground objects, the
    class appended as last column, plus im_info = (BEV rows, BEV columns, 1)."""
    fg = np.flatnonzero(entry['gt_classes'] != 0)
    cls = entry['gt_classes'][fg].astype(np.float32)[:, None]
    blob = lambda key: np.hstack([np.asarray(entry[key], np.float32)[fg], cls]).astype(np.float32)
    return {'gt_boxes': blob('boxes'), 'gt_boxes_bv': blob('boxes_bv'), 'gt_boxes_3d': blob('boxes_3D'),
            'gt_boxes_corners': blob('boxes_corners'),
            'im_info': np.array([[lidar_bv_shape[0], lidar_bv_shape[1], 1]], dtype=np.float32)}


class kitti_mv3d(object):
    """Dataset accessor with the reference class's public face (what train_net / test_net touch)."""

    def __init__(self, image_set, kitti_path):
        self._image_set, self._kitti_path = image_set, kitti_path
        self._data_path = os.path.join(kitti_path, 'object')
        self._classes = ('__background__', 'Car')
        self._class_to_ind = {c: i for i, c in enumerate(self._classes)}
        self._image_ext, self._lidar_ext = '.png', '.npy'
        index_file = os.path.join(kitti_path, 'ImageSets', image_set + '.txt')
        for p, what in ((kitti_path, 'KITTI path'), (self._data_path, 'Path'), (index_file, 'Path')):
            assert os.path.exists(p), '{} does not exist: {}'.format(what, p)
        with open(index_file) as f:
            self._image_index = f.read().split()
        self.name = 'kitti_mv3d_' + image_set
        self._roidb = None

    classes = property(lambda self: self._classes)
    num_classes = property(lambda self: len(self._classes))
    image_index = property(lambda self: self._image_index)
    num_images = property(lambda self: len(self._image_index))

    @property
    def roidb(self):
        if self._roidb is None:
            self._roidb = self.gt_roidb()
        return self._roidb

    def _dir(self, sub):
        return os.path.join(self._data_path, 'testing' if self._image_set == 'test' else 'training', sub)

    def _file(self, sub, i, ext):
        p = os.path.join(self._dir(sub), self._image_index[i] + ext)
        assert os.path.exists(p), 'Path does not exist: {}'.format(p)
        return p

    def image_path_at(self, i):
        return self._file('image_2', i, self._image_ext)

    def lidar_path_at(self, i):
        return self._file('lidar_bv', i, self._lidar_ext)

    def _load_kitti_calib(self, index):
        return load_kitti_calib(os.path.join(self._dir('calib'), index + '.txt'))

    def calib_at(self, i):
        return pack_calib(self._load_kitti_calib('%06d' % i))      # by POSITION, like the reference (kitti_mv3d.py:67)

    def _load_kitti_annotation(self, index):
        with open(os.path.join(self._data_path, 'training', 'label_2', index + '.txt')) as f:
            lines = f.readlines()
        return parse_kitti_labels(lines, self._load_kitti_calib(index)['Tr_velo2cam'], self._class_to_ind, self.num_classes)

    def gt_roidb(self):
        return [self._load_kitti_annotation(index) for index in self._image_index]

    def append_flipped_images(self):
        raise NotImplementedError("cfg.TRAIN.USE_FLIPPED: the reference's flip only mirrors the 2-D image boxes "
                                  "(lib/datasets/imdb.py:104-121) and would leave BEV / 3-D ground truth unflipped; not built")
