"""Known-answer tests pinning the CPU oracle to the reference kernels' documented semantics
(the reference itself ships no vectors for this path).  Every expected value is
hand-derived from the cited lines of /root/reference/multi_model/utils/pn2_utils/csrc."""
import numpy as np
import pytest
import torch

from oracle import pn2_ext_oracle as orc


def xyz(points):
    return torch.tensor(points, dtype=torch.float32).t().contiguous().unsqueeze(0)  # (1,3,N)


def test_fps_first_is_zero_and_farthest_next():
    # sampling_kernel.cu:65 -> index 0 first; then the point maximising the min-distance
    p = xyz([[0, 0, 0], [1, 0, 0], [3, 0, 0], [0.5, 0, 0]])
    assert orc.farthest_point_sample(p, 3).tolist() == [[0, 2, 1]]


def test_fps_tie_between_lanes_uses_tree_order():
    # N=4 -> block 16 (the switch's minimum, :148-165); lanes 1 and 2 hold equal maxima.  The tree
    # (:99-111) compares (t, t+8), (t, t+4), (t, t+2): lane 0 takes lane 2's value (0 < 1) while
    # lane 1 keeps its own; at offset 1 lane 0 (now holding point 2) is NOT < lane 1 (equal), so
    # point 2 wins although point 1 has the smaller lane/index.
    p = xyz([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0.1, 0, 0]])
    assert orc.farthest_point_sample(p, 2).tolist() == [[0, 2]]


def test_fps_tie_inside_one_lane_keeps_first():
    # block 16, N=18: points 1 and 17 share lane 1; strict > at :89-92 keeps the earlier one
    pts = [[0, 0, 0]] + [[0.01 * i, 0, 0] for i in range(1, 18)]
    pts[1] = [5, 0, 0]
    pts[17] = [-5, 0, 0]
    assert orc.farthest_point_sample(xyz(pts), 2).tolist() == [[0, 1]]


def test_fps_identical_points_repeat_current():
    # all distances 0: max_dist stays 0 with max_ind = cur_ind (:67-68) -> index 0 repeated
    p = xyz([[1, 1, 1]] * 5)
    assert orc.farthest_point_sample(p, 4).tolist() == [[0, 0, 0, 0]]


def test_fps_argument_checks():
    p = xyz([[0, 0, 0], [1, 0, 0]])
    with pytest.raises(RuntimeError):
        orc.farthest_point_sample(p, 0)
    with pytest.raises(RuntimeError):
        orc.farthest_point_sample(p, 3)


def test_ball_query_first_k_in_index_order_and_padding():
    # ball_query_kernel.cu:55-72: hits 1,2,4 (d2 < 1); K=5 -> [1,2,4,1,1], count 3
    p = xyz([[2, 0, 0], [0.1, 0, 0], [0, 0.2, 0], [0, 3, 0], [0, 0, 0.9]])
    c = xyz([[0, 0, 0]])
    idx, cnt = orc.ball_query(p, c, 1.0, 5)
    assert idx.tolist() == [[[1, 2, 4, 1, 1]]] and cnt.tolist() == [[3]]
    idx, cnt = orc.ball_query(p, c, 1.0, 2)     # early exit at K hits (:55)
    assert idx.tolist() == [[[1, 2]]] and cnt.tolist() == [[2]]


def test_ball_query_strict_radius_and_empty_ball():
    p = xyz([[1, 0, 0], [0.5, 0, 0]])
    c = xyz([[0, 0, 0], [10, 10, 10]])
    idx, cnt = orc.ball_query(p, c, 1.0, 3)     # d2 == r2 excluded (:61); empty ball stays zero (:107)
    assert idx.tolist() == [[[1, 1, 1], [0, 0, 0]]] and cnt.tolist() == [[1, 0]]


def test_three_nn_sorted_squared_distances_and_stable_ties():
    # interpolate_kernel.cu:59-69: ascending SQUARED distances; equal distances keep key order
    q = xyz([[0, 0, 0]])
    k = xyz([[2, 0, 0], [0, 1, 0], [0, 0, -1], [1, 0, 0], [0, 0, 3]])
    idx, d = orc.point_search(q, k, 3)
    assert idx.tolist() == [[[1, 2, 3]]] and d.tolist() == [[[1.0, 1.0, 1.0]]]
    with pytest.raises(RuntimeError):
        orc.point_search(q, k[:, :, :2], 3)
    with pytest.raises(RuntimeError):
        orc.point_search(q, k, 2)


def test_three_nn_exactly_three_keys():
    q = xyz([[0, 0, 0], [5, 0, 0]])
    k = xyz([[3, 0, 0], [1, 0, 0], [2, 0, 0]])
    idx, d = orc.point_search(q, k, 3)
    assert idx.tolist() == [[[1, 2, 0], [0, 2, 1]]]
    assert d.tolist() == [[[1.0, 4.0, 9.0], [4.0, 9.0, 16.0]]]


def test_group_and_gather_knn_match_torch_gather():
    # the reference's only self-check (functions/gather_knn.py:26-55): seed 1, B=2,C=4,N=5,K=3
    torch.manual_seed(1)
    B, C, N, K = 2, 4, 5, 3
    feat = torch.rand(B, C, N)
    idx = torch.randint(0, N, (B, N, K))
    want = torch.gather(feat.unsqueeze(2).expand(B, C, N, N), 3, idx.unsqueeze(1).expand(B, C, N, K))
    assert torch.equal(orc.gather_knn_forward(feat, idx), want)
    assert torch.equal(orc.group_points_forward(feat, idx), want)
    g = torch.ones(B, C, N, K)
    ref = torch.zeros(B, C, N).scatter_add_(2, idx.view(B, 1, -1).expand(B, C, -1), g.view(B, C, -1))
    assert torch.allclose(orc.gather_knn_backward(g, idx), ref)
    assert torch.allclose(orc.group_points_backward(g, idx, N), ref)


def test_interpolate_forward_backward():
    feat = torch.tensor([[[1.0, 2.0, 4.0], [10.0, 20.0, 40.0]]])          # (1,2,3)
    idx = torch.tensor([[[0, 1, 2], [2, 2, 0]]])                           # (1,2,3)
    w = torch.tensor([[[0.5, 0.25, 0.25], [0.5, 0.5, 0.0]]])
    out = orc.interpolate_forward(feat, idx, w)
    assert out.tolist() == [[[2.0, 4.0], [20.0, 40.0]]]
    g = torch.tensor([[[1.0, 2.0], [0.0, 4.0]]])
    gi = orc.interpolate_backward(g, idx, w, 3)
    assert gi.tolist() == [[[0.5, 0.25, 2.25], [0.0, 0.0, 4.0]]]


def test_non_contiguous_views_are_accepted():
    pc = torch.rand(2, 50, 6)
    view = pc.permute(0, 2, 1)[:, :3, :]               # the view ScoreNet passes (score_network.py:46)
    assert torch.equal(orc.farthest_point_sample(view, 10), orc.farthest_point_sample(view.contiguous(), 10))


def test_radius_mask_inclusive_on_sqrt():
    # get_regiondataset.py:293-294: sqrt(d2) <= R, inclusive
    pts = torch.tensor([[0.0, 0, 0, 0, 0, 0], [0.008, 0, 0, 0, 0, 0], [0.0081, 0, 0, 0, 0, 0]])
    m = orc.radius_mask(pts, pts[:1], float(np.float32(0.008)))
    assert m.tolist() == [[True, True, False]]
