"""Import-path alias: the reference's module path, served by regnet_for_3d_grasping_amd."""
from regnet_for_3d_grasping_amd.score_network import *  # noqa: F401,F403
from regnet_for_3d_grasping_amd.score_network import ScoreNetwork  # noqa: F401
