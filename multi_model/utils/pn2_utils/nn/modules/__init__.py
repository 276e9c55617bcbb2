"""Import-path alias: the reference's module path, served by regnet_for_3d_grasping_amd."""
from .conv import Conv1d, Conv2d  # noqa: F401
from .linear import FC  # noqa: F401
from .mlp import MLP, SharedMLP  # noqa: F401
