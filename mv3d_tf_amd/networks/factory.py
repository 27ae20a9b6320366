"""`get_network(name)`: interface of lib/networks/factory.py:23-33 for the two MV3D graphs."""
from .mv3d import MV3D


def get_network(name):
    """'MV3D_train' / 'MV3D_test' (the legacy VGGnet_* 2-D Faster-RCNN graphs are out of scope)."""
    if name.split('_')[0] != 'MV3D':
        raise KeyError('Unknown dataset: {}'.format(name))
    if name.split('_')[1] == 'test':
        return MV3D(phase="TEST")
    if name.split('_')[1] == 'train':
        return MV3D(phase="TRAIN")
    raise KeyError('Unknown dataset: {}'.format(name))


def list_networks():
    return ['MV3D_train', 'MV3D_test']
