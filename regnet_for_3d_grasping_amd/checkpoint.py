"""Checkpoint compatibility with the reference.

The reference saves WHOLE OBJECTS, ``torch.save(score_model, 'score_<epoch>.model')`` (train.py:467-468,
possibly wrapped in ``nn.DataParallel``), and restores them by unpickling, taking ``.state_dict()`` and
stripping the ``module.`` prefix (utils.py:59-90).  Such a pickle names its classes by the reference's import
paths (``multi_model.score_network.ScoreNetwork``, ``multi_model.utils.pn2_utils.nn.modules.conv.Conv2d`` ...).

Both directions work here:
  * loading: the repo root serves those import paths (``multi_model/``, aliases of this package), so the
    reference's pickles unpickle onto this package's classes; ``construct_scorenet`` / ``construct_rnet`` follow
    utils.py:59-90 (same arguments, same ``resume_num`` rule);
  * saving: ``save_model`` writes the same whole-object pickle with the classes named by the REFERENCE paths
    (``reference_class_paths``), so a checkpoint written here is laid out like the reference's own.
"""
import contextlib
import importlib

import torch

# class -> the module path the reference defines it in
_REFERENCE_MODULE = {
    "ScoreNetwork": "multi_model.score_network",
    "GripperRegionNetwork": "multi_model.gripper_region_network",
    "PointNet2Seg": "multi_model.utils.pointnet2",
    "PointNet2TwoStage": "multi_model.utils.pointnet2",
    "PointNet2Refine": "multi_model.utils.pointnet2",
    "FarthestPointSampler": "multi_model.utils.pn2_utils.modules",
    "QueryGrouper": "multi_model.utils.pn2_utils.modules",
    "EdgeQueryGrouper": "multi_model.utils.pn2_utils.modules",
    "FeatureInterpolator": "multi_model.utils.pn2_utils.modules",
    "EdgeFeatureInterpolator": "multi_model.utils.pn2_utils.modules",
    "PointNetSAModule": "multi_model.utils.pn2_utils.modules",
    "PointNetSAAvgModule": "multi_model.utils.pn2_utils.modules",
    "PointNetSAModuleMSG": "multi_model.utils.pn2_utils.modules",
    "EdgeSAModule": "multi_model.utils.pn2_utils.modules",
    "PointnetFPModule": "multi_model.utils.pn2_utils.modules",
    "EdgeFPModule": "multi_model.utils.pn2_utils.modules",
    "Conv1d": "multi_model.utils.pn2_utils.nn.modules.conv",
    "Conv2d": "multi_model.utils.pn2_utils.nn.modules.conv",
    "FC": "multi_model.utils.pn2_utils.nn.modules.linear",
    "MLP": "multi_model.utils.pn2_utils.nn.modules.mlp",
    "SharedMLP": "multi_model.utils.pn2_utils.nn.modules.mlp",
}


def _classes():
    for name, path in _REFERENCE_MODULE.items():
        cls = getattr(importlib.import_module(path), name)   # the alias module hands out this package's class
        yield cls, path


@contextlib.contextmanager
def reference_class_paths():
    """While active, this package's model classes pickle under the reference's import paths."""
    saved = []
    try:
        for cls, path in _classes():
            saved.append((cls, cls.__module__))
            cls.__module__ = path
        yield
    finally:
        for cls, module in saved:
            cls.__module__ = module


def save_model(model, path):
    """``torch.save(model, path)`` as train.py:467-468 does, class paths as in the reference."""
    # the fused forward caches packed weights on the modules (``_regnet_*`` attributes: this package's own classes and a
    # second copy of every weight): they are not part of the model and the reference could not unpickle them
    stripped = []
    for m in model.modules():
        for k in [k for k in m.__dict__ if k.startswith("_regnet_")]:
            stripped.append((m, k, m.__dict__.pop(k)))
    try:
        with reference_class_paths():
            torch.save(model, path)
    finally:
        for m, k, v in stripped:
            m.__dict__[k] = v


def load_state_dict(path, map_location="cpu"):
    """Unpickle a whole-object checkpoint and return its ``state_dict`` with the DataParallel ``module.``
    prefix removed (utils.py:66-69,:86-88)."""
    obj = torch.load(path, map_location=map_location, weights_only=False)
    state = obj.state_dict() if hasattr(obj, "state_dict") else obj
    return {key.replace("module.", ""): value for key, value in state.items()}


def _resume_num(model_path):
    """utils.py:71 / :92: ``<dir>/score_<epoch>.model`` -> epoch + 1."""
    return 1 + int(model_path.split("/")[-1].split("_")[1].split(".model")[0])


def construct_scorenet(load_flag, obj_class_num=2, model_path=None, gpu_num=0, map_location=None):
    """utils.py:59-72.  ``map_location`` defaults to the reference's ``cuda:<gpu_num>``."""
    from .score_network import ScoreNetwork
    score_model = ScoreNetwork(training=True, k_obj=obj_class_num)
    resume_num = 0
    if load_flag and model_path != "" and model_path is not None:
        where = map_location if map_location is not None else "cuda:{}".format(gpu_num)
        score_model.load_state_dict(load_state_dict(model_path, where))
        resume_num = _resume_num(model_path)
    return score_model, resume_num


def construct_rnet(load_flag, training_refine, group_num, gripper_num, grasp_score_threshold, depth, reg_channel,
                   model_path=None, gpu_num=0, map_location=None):
    """utils.py:74-93: checkpoint keys UPDATE the fresh model's state (a stage-2 checkpoint without the refine
    head still loads)."""
    from .gripper_region_network import GripperRegionNetwork
    region_model = GripperRegionNetwork(training=training_refine, group_num=group_num, gripper_num=gripper_num,
                                        grasp_score_threshold=grasp_score_threshold, radius=depth,
                                        reg_channel=reg_channel)
    resume_num = 0
    if load_flag and model_path != "" and model_path is not None:
        where = map_location if map_location is not None else "cuda:{}".format(gpu_num)
        cur = region_model.state_dict()
        cur.update(load_state_dict(model_path, where))
        region_model.load_state_dict(cur)
        resume_num = _resume_num(model_path)
    return region_model, resume_num
