from .fpn import FPN  # noqa: F401
