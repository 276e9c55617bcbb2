"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/regnet_hip.h declares; the Python binding rejects CPU tensors like the reference's
CHECK_CUDA; top-level import paths of the reference resolve to this implementation."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "regnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(regnet_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from regnet_for_3d_grasping_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), "libregnet_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "binding has no signature for %s" % name
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib.regnet_abi_version() >= 1
    assert b"gfx950" in _lib.lib.regnet_build_info()
    assert _lib.lib.regnet_strerror(-1).decode().startswith("shape")


def test_argument_checks_without_gpu():
    from regnet_for_3d_grasping_amd import _lib
    L = _lib.lib
    # validation happens before any launch, so these are safe without a device
    assert L.regnet_fps_f32(None, 0, 0, 0, 1, 10, 0, None, None, None) == -1      # M <= 0
    assert L.regnet_fps_f32(None, 0, 0, 0, 1, 10, 11, None, None, None) == -1     # N < M
    assert L.regnet_fps_f32(None, 0, 0, 0, 0, 10, 5, None, None, None) == 0       # empty batch
    assert L.regnet_fps_f32(None, 0, 0, 0, 1, 10, 5, None, None, None) == -2      # null pointers
    assert L.regnet_three_nn_f32(None, 0, 0, 0, None, 0, 0, 0, 1, 5, 2, None, None, None) == -1
    assert L.regnet_ball_query_f32(None, 0, 0, 0, None, 0, 0, 0, 1, 5, 5, 0.1, 0, None, None, None) == -1
    assert L.regnet_fps_workspace_bytes(4, 25600, 64) == 0                  # short run: register-resident kernel
    assert L.regnet_fps_workspace_bytes(4, 4096, 2048) == 0
    assert L.regnet_fps_workspace_bytes(4, 5120, 1024) == 4 * 5120 * 4         # level-2 shape: the cluster kernel as well
    assert L.regnet_fps_workspace_bytes(4, 25600, 5120) == 4 * 25600 * 4       # long run: the Morton permutation
    assert L.regnet_fps_f32(1, 3 * 25600, 25600, 1, 1, 25600, 5120, 1, None, None) == -2   # ... which must be provided
    # beyond 25 600 points: B x N words (permutation / running distances) + the cooperating workgroups' exchange area
    # ... + the launch's status word (a cooperating workgroup that lost its partner flags it instead of sampling on)
    assert L.regnet_fps_workspace_bytes(4, 51200, 5120) == 4 * 51200 * 4 + 4 * (2 * 4 * 64 * 8 * 4 + 256) + 256
    assert L.regnet_fps_status_offset_bytes(4, 51200, 5120) == 4 * 51200 * 4 + 4 * (2 * 4 * 64 * 8 * 4 + 256)
    assert L.regnet_fps_status_offset_bytes(4, 25600, 5120) == -1          # one workgroup per scene: nothing to lose
    assert L.regnet_fps_status_offset_bytes(4, 51200, 64) == -1            # short run: not the cooperative cluster kernel
    with pytest.raises(RuntimeError):
        _lib.check(-1, "x")


def test_binding_rejects_cpu_tensors_like_check_cuda():
    from regnet_for_3d_grasping_amd import dgcnn_ext, pn2_ext, region_ops
    x = torch.zeros(1, 3, 8)
    idx = torch.zeros(1, 4, 2, dtype=torch.int64)
    for call in (lambda: pn2_ext.farthest_point_sample(x, 2), lambda: pn2_ext.ball_query(x, x, 0.1, 2),
                 lambda: pn2_ext.point_search(x, x, 3), lambda: pn2_ext.group_points_forward(x, idx),
                 lambda: pn2_ext.group_points_backward(torch.zeros(1, 3, 4, 2), idx, 8),
                 lambda: pn2_ext.interpolate_forward(x, torch.zeros(1, 5, 3, dtype=torch.int64), torch.zeros(1, 5, 3)),
                 lambda: dgcnn_ext.gather_knn_forward(x, idx),
                 lambda: region_ops.radius_candidates(torch.zeros(1, 8, 6), torch.zeros(1, 2, 6), 0.1)):
        with pytest.raises(RuntimeError, match="CUDA tensor"):
            call()


def test_reference_import_paths_resolve_here():
    import importlib
    for name in ("pn2_ext", "dgcnn_ext", "multi_model.utils.pn2_utils.function",
                 "multi_model.utils.pn2_utils.modules", "multi_model.utils.pn2_utils.nn",
                 "multi_model.utils.pointnet2", "multi_model.score_network", "multi_model.gripper_region_network",
                 "dataset_utils.get_regiondataset", "dataset_utils.scoredataset", "dataset_utils.eval_score.eval"):
        mod = importlib.import_module(name)
        assert os.path.abspath(mod.__file__).startswith(REPO), name
    from multi_model.score_network import ScoreNetwork
    from multi_model.utils.pn2_utils import function as _F
    assert ScoreNetwork.__name__ == "ScoreNetwork" and callable(_F.farthest_point_sample)
    from dataset_utils.eval_score.eval import eval_test, eval_validate   # test.py:17
    assert callable(eval_test) and callable(eval_validate)
    for op in ("gather_points", "farthest_point_sample", "ball_query", "group_points", "search_nn_distance",
               "feature_interpolate"):
        assert hasattr(_F, op)
