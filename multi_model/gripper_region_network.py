"""Import-path alias: the reference's module path, served by regnet_for_3d_grasping_amd."""
from regnet_for_3d_grasping_amd.gripper_region_network import (  # noqa: F401
    GripperRegionNetwork, _enumerate_templates, compute_cos_sim, get_gripper_region_transform, gripper_frame)
