"""`RoIDataLayer`: epoch-shuffled frame iterator (interface of lib/roi_data_layer/layer.py:17-67).  The shuffle uses the
numpy GLOBAL RNG like the reference (`np.random.permutation(np.arange(n))`), so a seeded run visits the frames in the
reference's order; a new permutation is drawn when fewer than IMS_PER_BATCH frames are left (the HAS_RPN branch, :33-38)."""
import numpy as np

from ..fast_rcnn.config import cfg
from .minibatch_mv3d import get_minibatch


class RoIDataLayer(object):
    def __init__(self, roidb, num_classes):
        self._roidb, self._num_classes = roidb, num_classes
        self._shuffle_roidb_inds()

    def _shuffle_roidb_inds(self):
        self._perm = np.random.permutation(np.arange(len(self._roidb)))
        self._cur = 0

    def _get_next_minibatch_inds(self):
        step = cfg.TRAIN.IMS_PER_BATCH
        if self._cur + step >= len(self._roidb):
            self._shuffle_roidb_inds()
        inds = self._perm[self._cur:self._cur + step]
        self._cur += step
        return inds

    def forward(self):
        """blobs of the next frame"""
        return get_minibatch([self._roidb[i] for i in self._get_next_minibatch_inds()], self._num_classes)
