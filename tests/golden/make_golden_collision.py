"""AUTHORING-CONTAINER ONLY: generate tests/golden/s6_collision.npz from the reference's own view-cloud collision
filter (dataset_utils/eval_score/eval.py:eval_test -> EvalDataTest.run_collision_view).

The reference classes import open3d and transforms3d, which the image lacks.  Neither influences what eval_test /
eval_validate return for these inputs: transforms3d is only referenced by an unused helper
(evaluation_data_generator.py:40-43); open3d estimates normals of the VIEW cloud (:77-80, :261-262) whose transformed
copy (:199, :437) is never read, builds kd-trees nobody queries (:260) and would estimate SCENE normals only when the
record carries none (torch_scene_point_cloud.py:13-19; the fixture's records carry ``scene_normal``).  They are therefore
replaced by inert stand-ins (a point-cloud object that stores its points and returns NaN normals, so any read of them
would poison the fixture), and the fixture
records exactly what the reference's code returned for seeded inputs.  Run:  python tests/golden/make_golden_collision.py
"""
import contextlib
import importlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE_ROOT = os.environ.get("REGNET_REFERENCE_ROOT", "/root/reference")


def _inert_open3d():
    o3d = types.ModuleType("open3d")

    class _Cloud:
        def __init__(self):
            self.points = np.zeros((0, 3))
            self.normals = np.full((0, 3), np.nan)

        def estimate_normals(self, **kw):
            # NaN, not zero: a code path that READ these stand-in normals would poison the fixture instead of
            # silently agreeing with it (the claim "never read on these paths" is thereby checked, not just stated)
            self.normals = np.full((len(self.points), 3), np.nan)

        def normalize_normals(self):
            pass

        def orient_normals_towards_camera_location(self, cam):
            pass

    o3d.geometry = types.SimpleNamespace(PointCloud=_Cloud, KDTreeSearchParamHybrid=lambda **kw: None,
                                         KDTreeFlann=lambda cloud: None)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a))
    o3d.visualization = types.SimpleNamespace(draw_geometries=lambda *a, **k: None)
    return o3d


def import_reference_eval():
    sys.path.insert(0, REPO_ROOT)
    from regnet_for_3d_grasping_amd import synthetic   # bind the product package before the reference shadows names
    sys.modules["open3d"] = _inert_open3d()
    t3d = types.ModuleType("transforms3d")
    t3d.quaternions = types.SimpleNamespace(axangle2quat=lambda *a, **k: np.zeros(4))
    sys.modules["transforms3d"] = t3d
    for name in [m for m in sys.modules if m.split(".")[0] == "dataset_utils"]:
        del sys.modules[name]
    # this repo ships a regular package named ``dataset_utils`` (import-path aliases), which would win over the
    # reference's namespace package whatever the path order: hide the repo root while the reference is imported
    hidden = [e for e in sys.path if e in ("", ".") or os.path.abspath(e) in (REPO_ROOT, HERE)]
    saved_path = list(sys.path)
    sys.path[:] = [REFERENCE_ROOT] + [e for e in sys.path if e not in hidden]
    try:
        mod = importlib.import_module("dataset_utils.eval_score.eval")
    finally:
        sys.path[:] = saved_path
    assert mod.__file__.startswith(REFERENCE_ROOT), mod.__file__
    return mod, synthetic


def main():
    mod, _ = import_reference_eval()
    sys.path.insert(0, os.path.dirname(HERE))
    import golden_util
    out = {}
    for i, c in enumerate(golden_util.COLLISION_CASES):
        pts, g = golden_util.collision_case(i)
        with contextlib.redirect_stdout(io.StringIO()):
            kept = mod.eval_test(pts, g, None, c["table_height"], c["depth"], c["width"], -1)
        kept = kept.numpy()
        # which input rows survived (rows are unique: the score column is continuous)
        idx = np.array([int(np.nonzero((g == row).all(1))[0][0]) for row in kept], dtype=np.int64)
        out["c%d_kept_index" % i] = idx
        out["c%d_kept" % i] = kept
        print("case %d: %d of %d grasps kept" % (i, len(idx), len(g)))
    # validation flavour: eval_validate = view filter + scene filter + antipodal score (scene normals from the record)
    for i, c in enumerate(golden_util.VALIDATE_CASES):
        data, g = golden_util.validate_case(i)
        with contextlib.redirect_stdout(io.StringIO()):
            vgr, score, n_view, g_view, g_scene = mod.eval_validate(data, g, c["view_num"], c["table_height"], c["depth"],
                                                                    c["width"], -1)
        out["v%d_vgr" % i] = np.int64(vgr)
        out["v%d_score" % i] = np.float64(score)
        out["v%d_n_view" % i] = np.int64(n_view)
        out["v%d_view" % i] = g_view.numpy()
        out["v%d_scene" % i] = g_scene.numpy()
        print("validate case %d: %d grasps, %d without view collision, vgr %d, score %.6f" % (i, len(g), n_view, vgr, score))
    np.savez_compressed(os.path.join(HERE, "s6_collision.npz"), **out)


if __name__ == "__main__":
    main()
