"""Import-path alias: the reference's module path, served by regnet_for_3d_grasping_amd."""
from regnet_for_3d_grasping_amd.scoredataset import ScoreDataset  # noqa: F401
