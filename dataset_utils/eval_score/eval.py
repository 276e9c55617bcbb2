"""Import-path alias: the reference's module path (dataset_utils/eval_score/eval.py), served by regnet_for_3d_grasping_amd."""
from regnet_for_3d_grasping_amd.eval_collision import eval_test  # noqa: F401


def eval_validate(*args, **kwargs):
    raise NotImplementedError("eval_validate (antipodal scoring against the ground-truth scene; open3d normals) is outside "
                              "this package's scope -- see DESIGN.md")
