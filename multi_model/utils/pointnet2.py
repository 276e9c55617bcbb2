"""Import-path alias: the reference's module path, served by regnet_for_3d_grasping_amd."""
from regnet_for_3d_grasping_amd.pointnet2 import PointNet2Refine, PointNet2Seg, PointNet2TwoStage  # noqa: F401
