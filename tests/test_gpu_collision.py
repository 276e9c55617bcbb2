"""View-cloud collision filter on the GPU (csrc/region.hip:grasp_collision_kernel through the C ABI,
regnet_for_3d_grasping_amd/eval_collision.py) against the CPU oracle and the reference-generated fixture."""
import numpy as np
import pytest
import torch

from . import golden_util
from oracle import collision_oracle as co

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("case", [0, 1])
def test_counts_are_bit_exact_vs_oracle_and_filter_matches_reference_fixture(case):
    from regnet_for_3d_grasping_amd import eval_collision as ec
    c = golden_util.COLLISION_CASES[case]
    pts, g = golden_util.collision_case(case)
    frame, center = co.grasp_frames(torch.from_numpy(g))
    T = co.global_to_local(frame, center)
    want = co.collision_counts(pts, T.numpy(), c["depth"], c["width"])
    got = ec.collision_counts(torch.from_numpy(pts).to(DEV), T.to(DEV), c["depth"], c["width"]).cpu().numpy()
    assert np.array_equal(got, want)                                   # integer work: exact
    # the whole mirror (frames computed on the GPU with the reference's torch expressions) against what the
    # reference's own eval_test returned for these inputs
    fx = golden_util.load("s6_collision.npz")
    kept = ec.eval_test(pts, g, None, c["table_height"], c["depth"], c["width"], 0)
    assert kept.is_cuda and np.array_equal(kept.cpu().numpy(), fx["c%d_kept" % case])


def test_strided_points_and_edge_cases():
    from regnet_for_3d_grasping_amd import eval_collision as ec
    pts, g = golden_util.collision_case(0)
    pc6 = torch.zeros(len(pts), 6, device=DEV)
    pc6[:, :3] = torch.from_numpy(pts).to(DEV)
    frame, center = co.grasp_frames(torch.from_numpy(g))
    T = co.global_to_local(frame, center).to(DEV)
    a = ec.collision_counts(pc6[:, :3], T, 0.06, 0.08)                 # a (N,3) view of (N,6) rows
    b = ec.collision_counts(pc6[:, :3].contiguous(), T, 0.06, 0.08)
    t = ec.collision_counts(pc6[:, :3].t().contiguous().t(), T, 0.06, 0.08)   # coordinate-major storage
    assert torch.equal(a, b) and torch.equal(a, t)
    assert ec.eval_test(pts, np.zeros((0, 8), np.float32), None, 0.75, 0.06, 0.08, 0).shape == (0, 8)
    empty = ec.collision_counts(torch.zeros(0, 3, device=DEV), T[:5], 0.06, 0.08)
    assert empty.shape == (5, 4) and int(empty.abs().sum()) == 0
    with pytest.raises(RuntimeError):
        ec.collision_counts(torch.from_numpy(pts), T, 0.06, 0.08)      # CPU tensor: no CPU path
    with pytest.raises(RuntimeError):
        ec.eval_test(pts, g, None, 0.75, 0.06, 0.08, -1)


def test_inference_script_scale_4000_grasps_25600_points():
    """test.py's sizes: every refined grasp of 4000 centres against the 25 600-point view cloud, one launch."""
    from regnet_for_3d_grasping_amd import eval_collision as ec, synthetic
    pts = synthetic.make_scene(4300, 25600)[:, :3].astype(np.float32)
    rng = np.random.default_rng(9)
    g = np.zeros((4000, 8), dtype=np.float32)
    g[:, :3] = pts[rng.integers(0, len(pts), 4000)] + rng.normal(0, 0.01, (4000, 3)).astype(np.float32)
    g[:, 2] += rng.uniform(0, 0.08, 4000).astype(np.float32)
    g[:, 3:6] = rng.normal(size=(4000, 3)).astype(np.float32)
    g[:, 6] = rng.uniform(-1.2, 1.2, 4000).astype(np.float32)
    frame, center = co.grasp_frames(torch.from_numpy(g))
    T = co.global_to_local(frame, center)
    got = ec.collision_counts(torch.from_numpy(pts).to(DEV), T.to(DEV), 0.06, 0.08).cpu().numpy()
    rows = rng.choice(4000, 256, replace=False)                         # the oracle on a sample of the grasps
    assert np.array_equal(got[rows], co.collision_counts(pts, T.numpy()[rows], 0.06, 0.08))
    keep = co.accept(got, frame.numpy(), center.numpy(), 0.75, 0.06)
    out = ec.eval_test(torch.from_numpy(pts).to(DEV), torch.from_numpy(g).to(DEV), None, 0.75, 0.06, 0.08, 0)
    # frames computed on the GPU may differ from the CPU's in the last bit (sin / cos): allow a grasp or two whose
    # decisive point sits within an ulp of a box face
    assert abs(int(keep.sum()) - out.shape[0]) <= 2 and 0 < out.shape[0] < 4000


@pytest.mark.parametrize("case", [0, 1])
def test_eval_validate_matches_reference_fixture_and_oracle(case):
    """View filter + scene filter + antipodal score (EvalDataValidate.run_collision) against what the reference's own
    eval_validate returned, and the per-grasp statistics against the oracle."""
    from regnet_for_3d_grasping_amd import eval_collision as ec
    c = golden_util.VALIDATE_CASES[case]
    data, g = golden_util.validate_case(case)
    fx = golden_util.load("s6_collision.npz")
    vgr, score, n_view, g_view, g_scene = ec.eval_validate(data, g, c["view_num"], c["table_height"], c["depth"], c["width"], 0)
    assert vgr == int(fx["v%d_vgr" % case]) and n_view == int(fx["v%d_n_view" % case])
    assert np.array_equal(g_view.cpu().numpy(), fx["v%d_view" % case])
    assert np.array_equal(g_scene.cpu().numpy(), fx["v%d_scene" % case])
    assert abs(score - float(fx["v%d_score" % case])) <= 1e-5 * max(1.0, abs(score))
    # kernel statistics vs the oracle on identical matrices: counts exact, scores to 1e-6
    frame, center = co.grasp_frames(torch.from_numpy(g))
    T = co.global_to_local(frame, center)
    scene = torch.from_numpy(data["scene_cloud"]).to(DEV)
    nrm = torch.from_numpy(data["scene_normal"]).to(DEV)
    want = co.collision_counts(data["scene_cloud"], T.numpy(), c["depth"], c["width"])
    got = ec.collision_counts(scene, T.to(DEV), c["depth"], c["width"]).cpu().numpy()
    assert np.array_equal(got, want)
    rows = np.nonzero(want[:, 3] > 0)[0]
    s_want = co.antipodal_scores(data["scene_cloud"], data["scene_normal"], T.numpy()[rows], c["depth"], c["width"])
    s_got = ec.antipodal_scores(scene, nrm, T[rows].to(DEV), c["depth"], c["width"]).cpu().numpy()
    np.testing.assert_allclose(s_got, s_want, rtol=2e-6, atol=1e-7)
    # one depth per grasp (the reference's tensor-depth branch, :428-430) == the scalar when they are all equal
    per = torch.full((len(g),), float(c["depth"]))
    assert np.array_equal(ec.collision_counts(scene, T.to(DEV), per, c["width"]).cpu().numpy(), want)
    d2 = torch.linspace(0.03, 0.07, len(g))
    assert np.array_equal(ec.collision_counts(scene, T.to(DEV), d2, c["width"]).cpu().numpy(),
                          co.collision_counts(data["scene_cloud"], T.numpy(), d2.numpy(), c["width"]))


# ---- normal estimation (eval_utils/pointcloud.py:27-43) ------------------------------------------------------------------
def _check_normals(pts, camera=(0.0, 0.0, 0.0), radius=0.01, max_nn=30, min_well=0.5):
    from regnet_for_3d_grasping_amd import eval_collision as ec
    want, cnt_want, gap = co.estimate_normals(pts, camera, radius, max_nn, return_gap=True)
    got, cnt = ec.estimate_normals(torch.from_numpy(pts).to(DEV), camera, radius, max_nn, return_count=True)
    got, cnt = got.cpu().numpy().astype(np.float64), cnt.cpu().numpy()
    assert np.array_equal(cnt, cnt_want)                                      # same neighbourhoods (exact arithmetic)
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-6)
    assert np.all(np.einsum("ij,ij->i", got, np.asarray(camera)[None] - pts.astype(np.float64)) >= -1e-7)
    dot = np.einsum("ij,ij->i", got, want)
    # the eigenvector is determined up to (rounding of the covariance) / (relative eigenvalue gap): the oracle forms the
    # covariance from raw second moments of absolute coordinates like open3d (~1e-11 relative), the kernel from
    # coordinates relative to the query point
    well = gap > 1e-3
    assert well.mean() >= min_well
    assert np.all(dot[well] > 1 - 1e-6), float(dot[well].min())
    fixed = cnt_want < 3
    assert np.array_equal(got[fixed], want[fixed])
    return cnt_want


def test_normals_match_oracle_on_scene():
    from regnet_for_3d_grasping_amd import synthetic
    scene = synthetic.make_scene(4400, 6000)[:, :3].astype(np.float32)
    cnt = _check_normals(scene, radius=0.02)
    assert cnt.max() == 30 and cnt.min() < 30                                 # both the capped and the uncapped branch
    _check_normals(scene[:3000], camera=(0.1, -0.2, 2.0), radius=0.03, max_nn=64)
    _check_normals(scene[:2500], min_well=0.0)                                      # the reference's radius: sparse, many lone points


def test_normals_known_answers_and_edges():
    from regnet_for_3d_grasping_amd import eval_collision as ec
    rng = np.random.default_rng(11)
    nrm = np.array([0.2, -0.3, 1.0]); nrm /= np.linalg.norm(nrm)
    a = np.cross(nrm, [1.0, 0.3, -0.2]); a /= np.linalg.norm(a)
    b = np.cross(nrm, a)
    uv = rng.uniform(-0.04, 0.04, (4000, 2))
    plane = (0.6 * nrm + uv[:, :1] * a + uv[:, 1:] * b).astype(np.float32)
    n, cnt = ec.estimate_normals(torch.from_numpy(plane).to(DEV), return_count=True)
    n, cnt = n.cpu().numpy().astype(np.float64), cnt.cpu().numpy()
    assert cnt.max() == 30 and np.all(np.abs(n[cnt >= 3] @ nrm + 1.0) < 1e-5)
    # a regular lattice: ties in distance everywhere -> the (distance, index) ranking must still match the oracle
    g = np.stack(np.meshgrid(np.arange(40), np.arange(40), indexing="ij"), -1).reshape(-1, 2) * 0.004
    lattice = np.concatenate([g, np.full((len(g), 1), 0.7)], 1).astype(np.float32)
    _check_normals(lattice, radius=0.015)
    # lone points, duplicates, empty and single-point clouds
    lone = np.array([[0.0, 0.0, 0.5], [0.3, 0.0, -0.5], [0.3, 0.005, -0.5]], dtype=np.float32)
    assert np.array_equal(ec.estimate_normals(torch.from_numpy(lone).to(DEV)).cpu().numpy(), [[0, 0, -1], [0, 0, 1], [0, 0, 1]])
    dup = np.repeat(np.array([[0.1, 0.1, 0.6]], dtype=np.float32), 70, axis=0)
    _check_normals(dup, min_well=0.0)
    assert np.array_equal(ec.estimate_normals(torch.from_numpy(dup).to(DEV)).cpu().numpy(), co.estimate_normals(dup)[0])
    assert ec.estimate_normals(torch.zeros((0, 3), device=DEV)).shape == (0, 3)
    with pytest.raises(RuntimeError):
        ec.estimate_normals(torch.zeros((4, 3)))                                # CPU tensors are refused, no fallback


def test_eval_validate_estimates_missing_scene_normals():
    """A validation record without scene_normal (torch_scene_point_cloud.py:17-19): the result equals the one obtained
    with the estimated normals passed in, and the score equals the oracle's on its own estimated normals."""
    from regnet_for_3d_grasping_amd import eval_collision as ec
    c = golden_util.VALIDATE_CASES[0]
    data, g = golden_util.validate_case(0)
    data = {k: (v[:6000] if k.startswith("scene") else v) for k, v in data.items()}
    bare = {k: v for k, v in data.items() if k != "scene_normal"}
    est = ec.estimate_normals(torch.from_numpy(bare["scene_cloud"]).to(DEV))
    a = ec.eval_validate(bare, g, c["view_num"], c["table_height"], c["depth"], c["width"], 0)
    b = ec.eval_validate(dict(bare, scene_normal=est.cpu().numpy()), g, c["view_num"], c["table_height"], c["depth"], c["width"], 0)
    assert a[0] == b[0] and a[2] == b[2] and abs(a[1] - b[1]) <= 1e-6 * max(1.0, abs(b[1])) and torch.equal(a[4], b[4])
    n_or, _ = co.estimate_normals(bare["scene_cloud"])
    want = co.eval_validate(dict(bare, scene_normal=n_or.astype(np.float32)), torch.from_numpy(g), c["view_num"], c["table_height"],
                            c["depth"], c["width"])
    assert a[0] == want[0] and a[2] == want[2]
    assert abs(a[1] - want[1]) <= 1e-4 * max(1.0, abs(want[1]))
