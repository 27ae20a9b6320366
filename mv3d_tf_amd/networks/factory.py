"""`get_network(name)`: interface of lib/networks/factory.py:23-33 for the two MV3D graphs."""
from .mv3d import MV3D


def get_network(name):
    """'MV3D_train' / 'MV3D_test' (the legacy VGGnet_* 2-D Faster-RCNN graphs are out of scope); 'MV3D_train_3view' /
    'MV3D_test_3view' add the front-view tower the reference leaves a TODO (networks/mv3d.py)."""
    parts = name.split('_')
    if parts[0] != 'MV3D' or len(parts) < 2:
        raise KeyError('Unknown dataset: {}'.format(name))
    views = 3 if parts[2:] == ['3view'] else 2
    if parts[2:] not in ([], ['3view']):
        raise KeyError('Unknown dataset: {}'.format(name))
    if parts[1] == 'test':
        return MV3D(phase="TEST", views=views)
    if parts[1] == 'train':
        return MV3D(phase="TRAIN", views=views)
    raise KeyError('Unknown dataset: {}'.format(name))


def list_networks():
    return ['MV3D_train', 'MV3D_test', 'MV3D_train_3view', 'MV3D_test_3view']
