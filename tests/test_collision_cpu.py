"""View-cloud collision filter (the part test.py uses): the oracle restatement against the fixture
the reference's own ``eval_test`` produced (tests/golden/make_golden_collision.py), plus hand-derived known answers of
every rule of EvalDataTest.finger_hand_view."""
import numpy as np
import torch

from . import golden_util
from oracle import collision_oracle as co


def test_oracle_reproduces_reference_eval_test_fixture():
    fx = golden_util.load("s6_collision.npz")
    for i, c in enumerate(golden_util.COLLISION_CASES):
        pts, g = golden_util.collision_case(i)
        kept = co.eval_test(pts, torch.from_numpy(g), None, c["table_height"], c["depth"], c["width"])
        assert np.array_equal(kept.numpy(), fx["c%d_kept" % i])
        assert np.array_equal(g[fx["c%d_kept_index" % i]], fx["c%d_kept" % i])
        assert 0 < len(kept) < len(g)          # the fixture exercises both outcomes


def _grasp(center, axis_y=(0, 1, 0), angle=0.0):
    return torch.tensor([[*center, *axis_y, angle, 1.0]], dtype=torch.float32)


def test_frame_conventions_and_fallbacks():
    # axis_y = +y, angle 0: axis_x = (y1, -y0, 0) = (1, 0, 0); approach = x; minor normal = x cross y = z
    frame, center = co.grasp_frames(_grasp((0.1, 0.2, 0.9)))
    assert torch.allclose(frame[0], torch.eye(3)) and torch.equal(center[0], torch.tensor([0.1, 0.2, 0.9]))
    # zero axis_y falls back to (0,1,0) (:139); axis_y along z gives a zero axis_x -> (1,0,0) (:144)
    f0, _ = co.grasp_frames(_grasp((0, 0, 1), axis_y=(0, 0, 0)))
    assert torch.allclose(f0[0], torch.eye(3))
    f1, _ = co.grasp_frames(_grasp((0, 0, 1), axis_y=(0, 0, 2)))
    assert torch.allclose(f1[0][:, 1], torch.tensor([0.0, 0.0, 1.0])) and torch.allclose(f1[0][:, 0], torch.tensor([1.0, 0.0, 0.0]))
    # the local transform maps the centre to the origin
    T = co.global_to_local(frame, center)
    assert torch.allclose(T[0] @ torch.tensor([0.1, 0.2, 0.9, 1.0]), torch.tensor([0.0, 0.0, 0.0, 1.0]), atol=1e-7)


def test_rules_known_answers():
    # grasp at the origin with the identity frame: local coordinates ARE the point coordinates, so the faces of the
    # boxes are hit exactly (float32(0.005) == float32(0.005)); the table is moved out of the way
    depth, width, th = 0.06, 0.08, -1.0
    c = np.array([0.0, 0.0, 0.0], dtype=np.float32)
    slab = np.stack([np.linspace(0.001, 0.05, 20), np.zeros(20), np.full(20, 0.02)], 1).astype(np.float32) + c
    g = _grasp(c)

    def run(extra):
        pts = np.concatenate([slab, np.asarray(extra, dtype=np.float32).reshape(-1, 3) + c], 0)
        frame, center = co.grasp_frames(g)
        counts = co.collision_counts(pts, co.global_to_local(frame, center).numpy(), depth, width)
        return counts[0], bool(co.accept(counts, frame.numpy(), center.numpy(), th, depth)[0])

    assert run(np.zeros((0, 3)))[1]                                             # 20 slab points >= 16, no collision
    assert run(np.zeros((0, 3)))[0].tolist() == [20, 0, 0, 0]                  # the slab points sit above the hand plane
    assert not run([[-0.01, 0.0, 0.0]])[1]                                      # a point behind the hand (x < 0)
    assert run([[-0.01, 0.0, 0.0]])[0].tolist() == [21, 1, 0, 1]
    assert run([[0.0, 0.0, 0.0]])[1]                                            # x == 0 is not behind (strict <)
    assert not run([[0.02, 0.045, 0.0]])[1]                                     # inside the left finger
    assert run([[0.02, 0.04, 0.0]])[1] and run([[0.02, 0.05, 0.0]])[1]          # finger faces are exclusive
    assert run([[0.02, 0.045, 0.005]])[1]                                       # |z| == half thickness: outside
    assert not run([[0.02, -0.045, 0.004]])[1]                                  # right finger
    assert run([[-0.06, 0.0, 0.0]])[0][0] == 20                                 # x == -bottom: outside the slab
    assert run([[0.06, 0.045, 0.0]])[1]                                         # x == depth: outside the slab
    # fewer than 16 points in the slab -> rejected (:203)
    frame, center = co.grasp_frames(g)
    counts = co.collision_counts(slab[:15], co.global_to_local(frame, center).numpy(), depth, width)
    assert counts[0, 0] == 15 and not co.accept(counts, frame.numpy(), center.numpy(), th, depth)[0]
    # finger tips below table + 5 mm -> rejected before any scan (:195): approach pointing down from 0.81
    low = _grasp((0.0, 0.0, 0.81), angle=-float(np.pi / 2))   # approach = cos t * x + sin t * z = -z
    frame, center = co.grasp_frames(low)
    assert frame[0, 2, 0] < -0.99
    big = np.array([[20, 0, 0, 20]], dtype=np.int32)
    assert not co.accept(big, frame.numpy(), center.numpy(), 0.75, depth)[0]
    assert co.accept(big, frame.numpy(), center.numpy(), 0.70, depth)[0]
    assert co.eval_test(slab, torch.zeros(0, 8), None, th, depth, width).shape == (0, 8)


def test_oracle_reproduces_reference_eval_validate_fixture():
    fx = golden_util.load("s6_collision.npz")
    for i, c in enumerate(golden_util.VALIDATE_CASES):
        data, g = golden_util.validate_case(i)
        vgr, score, n_view, g_view, g_scene = co.eval_validate(data, torch.from_numpy(g), c["view_num"], c["table_height"],
                                                               c["depth"], c["width"])
        assert vgr == int(fx["v%d_vgr" % i]) and n_view == int(fx["v%d_n_view" % i]) and vgr > 0
        assert np.array_equal(g_view.numpy(), fx["v%d_view" % i]) and np.array_equal(g_scene.numpy(), fx["v%d_scene" % i])
        assert abs(score - float(fx["v%d_score" % i])) <= 1e-5 * max(1.0, abs(score))


def test_antipodal_score_known_answer():
    # identity frame at the origin; closing region |y| < 0.04, |z| < 0.005, -0.06 < x < 0.06
    pts = np.array([[0.01, 0.030, 0.0], [0.01, 0.029, 0.0], [0.01, 0.020, 0.0],      # left: y_max = 0.03
                    [0.01, -0.030, 0.0], [0.01, -0.0295, 0.0], [0.01, 0.0, 0.0],     # right: y_min = -0.03
                    [0.01, 0.030, 0.02]], dtype=np.float32)                            # outside (|z|)
    nrm = np.array([[0, 1.0, 0], [0, -0.5, 0], [0, 0.9, 0], [0, 0.25, 0], [0, 0.75, 0], [0, 0.1, 0], [0, 9.0, 0]], np.float32)
    T = np.eye(4, dtype=np.float32)[None]
    # d = min((0.03 + 0.03) / 3, 0.005) = 0.005: left = {0.030, 0.029} -> mean(1, 0.5); right = {-0.030, -0.0295} -> mean(.25, .75)
    s = co.antipodal_scores(pts, nrm, T, 0.06, 0.08)
    assert abs(float(s[0]) - 0.75 * 0.5) < 1e-6
    assert np.isnan(co.antipodal_scores(pts[-1:], nrm[-1:], T, 0.06, 0.08)[0])      # empty region


# ---- normal estimation (eval_utils/pointcloud.py:27-43; open3d restated, parity unpinned: known answers only) -----------
def _plane_cloud(rng, n, normal, offset, extent=0.04):
    normal = np.asarray(normal, dtype=np.float64) / np.linalg.norm(normal)
    a = np.cross(normal, [1.0, 0.3, -0.2]); a /= np.linalg.norm(a)
    b = np.cross(normal, a)
    uv = rng.uniform(-extent, extent, (n, 2))
    return (offset * normal + uv[:, :1] * a + uv[:, 1:] * b).astype(np.float32)


def test_normals_oracle_known_answers():
    rng = np.random.default_rng(7)
    # a tilted plane 0.6 m in front of the camera: every point with >= 3 neighbours gets the plane normal facing the origin
    nrm = np.array([0.2, -0.3, 1.0]); nrm /= np.linalg.norm(nrm)
    pts = _plane_cloud(rng, 1500, nrm, 0.6)
    n, cnt = co.estimate_normals(pts)
    ok = cnt >= 3
    assert ok.sum() > 1400 and cnt.max() == 30                     # capped at max_nn
    assert np.all(np.abs(n[ok] @ nrm + 1.0) < 1e-6)                # float32 coordinates: the plane is flat to ~1e-8 m
    assert np.all(np.einsum("ij,ij->i", n, -pts.astype(np.float64)) >= 0)
    # fewer than three neighbours: (0,0,1), flipped when the camera is below the point
    lone = np.array([[0.0, 0.0, 0.5], [0.3, 0.0, -0.5], [0.3, 0.005, -0.5]], dtype=np.float32)
    n, cnt = co.estimate_normals(lone)
    assert cnt.tolist() == [1, 2, 2]
    assert np.array_equal(n, [[0, 0, -1], [0, 0, 1], [0, 0, 1]])
    # a sphere of radius 5 cm seen from outside: radial normals (neighbourhood of 1 cm -> a few degrees of curvature)
    v = rng.normal(size=(6000, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    centre = np.array([0.0, 0.0, 0.7])
    sph = (centre + 0.05 * v).astype(np.float32)
    n, cnt = co.estimate_normals(sph, camera_pos=centre + 10 * (sph[0] - centre))
    ok = cnt >= 6
    cosang = np.abs(np.einsum("ij,ij->i", n[ok], v[ok]))
    assert ok.sum() > 5000 and np.percentile(cosang, 5) > 0.995
    # duplicates only: zero covariance -> the solver's first basis vector, still a unit vector facing the camera
    dup = np.repeat(np.array([[0.1, 0.1, 0.6]], dtype=np.float32), 5, axis=0)
    n, cnt = co.estimate_normals(dup)
    assert cnt.tolist() == [5] * 5 and np.allclose(np.linalg.norm(n, axis=1), 1.0)
    assert np.all(n @ -dup[0].astype(np.float64) >= 0)
