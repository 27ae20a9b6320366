from .factory import get_network  # noqa: F401
