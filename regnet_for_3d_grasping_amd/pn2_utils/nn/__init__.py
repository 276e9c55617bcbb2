"""Per-point shared-MLP building blocks (mirror of pn2_utils/nn)."""
from .blocks import FC, MLP, Conv1d, Conv2d, SharedMLP  # noqa: F401
from . import init  # noqa: F401
