"""`get_minibatch(roidb, num_classes)`: the blobs of one training frame (interface of
lib/roi_data_layer/minibatch_mv3d.py:17-76).

Keys and dtypes as the reference feeds them: image_data (1,H,W,3) f32 BGR minus PIXEL_MEANS,
lidar_bv_data (1,Hbv,Wbv,9), calib (4,12), gt_boxes / gt_boxes_bv (G,5), gt_boxes_3d (G,7), gt_boxes_corners (G,25),
im_info (1,3) = BEV size.  One numpy draw is kept for RNG-stream parity with the reference (`npr.randint` over
cfg.TRAIN.SCALES, :22-23, whose result the reference never uses either).  A roidb entry may carry the frame in memory
('image' (H,W,3) BGR array, 'lidar_bv' array) instead of 'image_path' / 'lidar_bv_path'; image files are read with
numpy (.npy) or PIL -- cv2 is not a dependency here."""
import numpy as np
import numpy.random as npr

from ..datasets.kitti_mv3d import gt_blobs
from ..fast_rcnn.config import cfg


def _read_image_bgr(path):
    if path.endswith(".npy"):
        return np.load(path)
    from PIL import Image                                    # what cv2.imread returns: uint8 BGR
    return np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]


def get_minibatch(roidb, num_classes):
    num_images = len(roidb)
    npr.randint(0, high=len(cfg.TRAIN.SCALES), size=num_images)          # RNG-stream parity (:22-23)
    assert cfg.TRAIN.BATCH_SIZE % num_images == 0, \
        'num_images ({}) must divide BATCH_SIZE ({})'.format(num_images, cfg.TRAIN.BATCH_SIZE)
    assert num_images == 1, "Single batch only"
    entry = roidb[0]
    image = entry['image'] if 'image' in entry else _read_image_bgr(entry['image_path'])
    image = image.astype(np.float32, copy=True) - cfg.PIXEL_MEANS
    bev = entry['lidar_bv'] if 'lidar_bv' in entry else np.load(entry['lidar_bv_path'])
    blobs = {'image_data': image[None].astype(np.float32), 'lidar_bv_data': np.asarray(bev)[None], 'calib': entry['calib']}
    blobs.update(gt_blobs(entry, np.asarray(bev).shape))
    return blobs
