"""Import-path alias: the reference's module path (dataset_utils/eval_score/eval.py), served by regnet_for_3d_grasping_amd."""
from regnet_for_3d_grasping_amd.eval_collision import eval_test, eval_validate  # noqa: F401
