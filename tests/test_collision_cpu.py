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
    assert run(np.zeros((0, 3)))[0].tolist() == [20, 0, 0]
    assert not run([[-0.01, 0.0, 0.0]])[1]                                      # a point behind the hand (x < 0)
    assert run([[-0.01, 0.0, 0.0]])[0].tolist() == [21, 1, 0]
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
    big = np.full((1, 3), 20, dtype=np.int32) * np.array([[1, 0, 0]], dtype=np.int32)
    assert not co.accept(big, frame.numpy(), center.numpy(), 0.75, depth)[0]
    assert co.accept(big, frame.numpy(), center.numpy(), 0.70, depth)[0]
    assert co.eval_test(slab, torch.zeros(0, 8), None, th, depth, width).shape == (0, 8)
