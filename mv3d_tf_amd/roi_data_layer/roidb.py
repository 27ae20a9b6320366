"""`prepare_roidb(imdb)`: enrich a dataset's roidb with what training reads (interface of
lib/roi_data_layer/roidb.py:15-60): file paths, the (4,12) calibration table, and per-object max overlap / class taken from
the one-hot `gt_overlaps`."""
import numpy as np


def prepare_roidb(imdb):
    roidb = imdb.roidb
    for i, entry in enumerate(roidb):
        entry['image_path'] = imdb.image_path_at(i)
        entry['lidar_bv_path'] = imdb.lidar_path_at(i)
        entry['calib'] = imdb.calib_at(i)
        dense = entry['gt_overlaps'].toarray() if hasattr(entry['gt_overlaps'], 'toarray') else np.asarray(entry['gt_overlaps'])
        entry['max_overlaps'] = dense.max(axis=1) if dense.size else np.zeros((0,), dense.dtype)
        entry['max_classes'] = dense.argmax(axis=1) if dense.size else np.zeros((0,), np.int64)
        # background rows have class 0, foreground rows a positive class (roidb.py:52-58)
        assert all(entry['max_classes'][entry['max_overlaps'] == 0] == 0)
        assert all(entry['max_classes'][entry['max_overlaps'] > 0] != 0)
    return roidb
