"""Operator API (mirror of multi_model/utils/pn2_utils/function.py:11-175).

Each op is a ``torch.autograd.Function`` whose forward/backward call the native extension
module ``pn2_ext`` -- here the ctypes binding of libregnet_hip.so.  Only feature tensors
receive gradients (function.py:103-107,167-172); sampling / query ops are index producers and
return ``None`` gradients (function.py:46-48,76-78,131-133).
"""
import torch
from torch.autograd import Function

from .. import pn2_ext


def gather_points(points, index):
    """points (B,C,N), index (B,M) -> (B,C,M): pick columns (function.py:11-26).  The PUBLIC operator: ``torch.gather``, like
    the reference -- an index outside [0, N) is a device-side assertion, not a silent zero."""
    b, c, _ = points.shape
    return torch.gather(points, 2, index[:, None, :].expand(b, c, index.size(1)))


def gather_sampled_points(points, index):
    """``gather_points`` for indices THIS package produced (furthest point sampling, ball query: always in range): one native
    launch (csrc/gather.hip: an out-of-range index would read as zero and only set the shared status word) instead of
    ATen's expanded-index gather.  Internal call sites only; callers' own indices go through ``gather_points``."""
    if (points.is_cuda and points.dtype == torch.float32 and index.dtype == torch.int64 and index.is_cuda
            and not (torch.is_grad_enabled() and points.requires_grad)):
        return pn2_ext.gather_points(points, index)
    return gather_points(points, index)


class FarthestPointSample(Function):
    """(B,3,N) xyz -> (B,M) int64 indices; first centroid is point 0 (function.py:29-51)."""

    @staticmethod
    def forward(ctx, points, num_centroids):
        return pn2_ext.farthest_point_sample(points, num_centroids)

    @staticmethod
    def backward(ctx, *grads):
        return None, None


class BallQuery(Function):
    """-> index (B,M,K) of the first K in-radius points, count (B,M) (function.py:54-81)."""

    @staticmethod
    def forward(ctx, points, centroids, radius, num_neighbours):
        index, count = pn2_ext.ball_query(points, centroids, radius, num_neighbours)
        return index, count

    @staticmethod
    def backward(ctx, *grads):
        return None, None, None, None


class GroupPoints(Function):
    """points (B,C,N), index (B,M,K) -> (B,C,M,K); backward scatter-adds (function.py:84-110)."""

    @staticmethod
    def forward(ctx, points, index):
        ctx.save_for_backward(index)
        ctx.num_points = points.size(2)
        return pn2_ext.group_points_forward(points, index)

    @staticmethod
    def backward(ctx, *grads):
        (index,) = ctx.saved_tensors
        return pn2_ext.group_points_backward(grads[0], index, ctx.num_points), None


class SearchNNDistance(Function):
    """query (B,3,N1), key (B,3,N2) -> index (B,N1,3), SQUARED distance (B,N1,3)
    (function.py:113-136)."""

    @staticmethod
    def forward(ctx, query_xyz, key_xyz, num_neighbors):
        index, distance = pn2_ext.point_search(query_xyz, key_xyz, num_neighbors)
        return index, distance

    @staticmethod
    def backward(ctx, *grads):
        return None, None, None


class FeatureInterpolate(Function):
    """feature (B,C,N2), index/weight (B,N1,3) -> (B,C,N1) (function.py:146-175)."""

    @staticmethod
    def forward(ctx, feature, index, weight):
        ctx.save_for_backward(index, weight)
        ctx.num_inst = feature.size(2)
        return pn2_ext.interpolate_forward(feature, index, weight)

    @staticmethod
    def backward(ctx, *grads):
        index, weight = ctx.saved_tensors
        return pn2_ext.interpolate_backward(grads[0], index, weight, ctx.num_inst), None, None


farthest_point_sample = FarthestPointSample.apply
ball_query = BallQuery.apply
group_points = GroupPoints.apply
search_nn_distance = SearchNNDistance.apply
feature_interpolate = FeatureInterpolate.apply
