"""Training data layer: interface of lib/roi_data_layer (layer.py:17-67, minibatch_mv3d.py:17-76)."""
from .layer import RoIDataLayer  # noqa: F401
from .minibatch_mv3d import get_minibatch  # noqa: F401
