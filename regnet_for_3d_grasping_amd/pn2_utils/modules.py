"""Set-abstraction / feature-propagation modules (mirror of
multi_model/utils/pn2_utils/modules.py:11-549).

Class names, constructor signatures, sub-module attribute names (``mlp``, ``sampler``,
``grouper``, ``interpolator``) and forward semantics follow the reference so that state_dicts
and model code are interchangeable.  REGNet instantiates only ``PointNetSAModule`` and
``PointnetFPModule`` (pointnet2.py:9); the Avg / MSG / Edge variants keep the API surface.

Tensor convention (reference): xyz ``(B,3,N)``, features ``(B,C,N)``, channel-first, fp32.
"""
import torch
from torch import nn

from . import function as _F
from .functions.gather_knn import gather_knn
from .nn import SharedMLP


class FarthestPointSampler(nn.Module):
    """xyz (B,3,N) -> (B,num_centroids) int64, under no_grad (modules.py:11-29)."""

    def __init__(self, num_centroids):
        super().__init__()
        self.num_centroids = num_centroids

    def forward(self, points):
        with torch.no_grad():
            return _F.farthest_point_sample(points, self.num_centroids)

    def extra_repr(self):
        return "num_centroids={:d}".format(self.num_centroids)


def _ball_group(radius, k, new_xyz, xyz, index=None):
    """Ball query + xyz grouping shared by the groupers: returns (index, centred group_xyz).
    group_xyz is (B,3,M,K) with the centroid subtracted in place (modules.py:41-46).
    ``index``: the ball-query result when it was computed ahead of time (a geometry plan)."""
    if index is None:
        with torch.no_grad():
            index, _ = _F.ball_query(xyz, new_xyz, radius, k)
    group_xyz = _F.group_points(xyz, index)
    group_xyz -= new_xyz.unsqueeze(-1)
    return index, group_xyz


class QueryGrouper(nn.Module):
    """Groups neighbours of each centroid; output channel order is [rel-xyz | feature]
    (modules.py:32-59)."""

    def __init__(self, radius, num_neighbours):
        super().__init__()
        assert radius > 0.0 and num_neighbours > 0
        self.radius, self.num_neighbours = radius, num_neighbours

    def forward(self, new_xyz, xyz, feature, use_xyz, index=None):
        index, group_xyz = _ball_group(self.radius, self.num_neighbours, new_xyz, xyz, index)
        if feature is None:
            return group_xyz, group_xyz
        group_feature = _F.group_points(feature, index)
        if use_xyz:
            group_feature = torch.cat([group_xyz, group_feature], dim=1)
        return group_feature, group_xyz

    def extra_repr(self):
        return "radius={}, num_neighbours={}".format(self.radius, self.num_neighbours)


class EdgeQueryGrouper(QueryGrouper):
    """EdgeConv flavour: appends (neighbour - centroid) features (modules.py:65-95)."""

    def forward(self, new_xyz, xyz, centroid_feature, feature, use_xyz):
        index, group_xyz = _ball_group(self.radius, self.num_neighbours, new_xyz, xyz)
        if feature is None:
            return group_xyz, group_xyz
        group_feature = _F.group_points(feature, index)
        parts = [group_feature, group_feature - centroid_feature.unsqueeze(-1)]
        if use_xyz:
            parts.insert(0, group_xyz)
        return torch.cat(parts, dim=1), group_xyz


def _three_nn_weights(dense_xyz, sparse_xyz, k, eps, geo=None):
    """3-NN indices + inverse-SQUARED-distance weights, normalised (modules.py:115-122).
    ``geo``: dict(idx, dist2) when the search was done ahead of time (a geometry plan)."""
    with torch.no_grad():
        if geo is not None:
            index, dist2 = geo["idx"], geo["dist2"]
        else:
            index, dist2 = _F.search_nn_distance(dense_xyz, sparse_xyz, k)
        inv = 1.0 / torch.clamp(dist2, min=eps)
        weight = inv / torch.sum(inv, dim=2, keepdim=True)
    return index, weight


class FeatureInterpolator(nn.Module):
    """Propagates sparse features to dense points; output is [interpolated | dense_feature]
    (modules.py:98-134)."""

    def __init__(self, num_neighbors, eps=1e-10):
        super().__init__()
        self.num_neighbors, self._eps = num_neighbors, eps

    def forward(self, dense_xyz, sparse_xyz, dense_feature, sparse_feature, geo=None):
        index, weight = _three_nn_weights(dense_xyz, sparse_xyz, self.num_neighbors, self._eps, geo)
        out = _F.feature_interpolate(sparse_feature, index, weight)
        if dense_feature is not None:
            out = torch.cat([out, dense_feature], dim=1)
        return out

    def extra_repr(self):
        return "num_neighbours={:d}, eps={}".format(self.num_neighbors, self._eps)


class EdgeFeatureInterpolator(FeatureInterpolator):
    """(B,C,N1,K) edge features: [interp | gathered - interp | dense] (modules.py:137-173)."""

    def forward(self, dense_xyz, sparse_xyz, dense_feature, sparse_feature):
        index, weight = _three_nn_weights(dense_xyz, sparse_xyz, self.num_neighbors, self._eps)
        with torch.no_grad():
            gathered = gather_knn(sparse_feature, index)
        k = self.num_neighbors
        interp = _F.feature_interpolate(sparse_feature, index, weight).unsqueeze(-1).expand(-1, -1, -1, k)
        parts = [interp, gathered - interp]
        if dense_feature is not None:
            parts.append(dense_feature.unsqueeze(-1).expand(-1, -1, -1, k))
        return torch.cat(parts, dim=1)


class _SetAbstraction(nn.Module):
    """Shared body of the single-scale SA variants: sample -> group -> SharedMLP -> reduce."""

    _grouper_cls = QueryGrouper

    def _build(self, in_channels, mlp_channels, num_centroids, radius, num_neighbours, use_xyz, mlp_in):
        self.in_channels = in_channels
        self.out_channels = mlp_channels[-1]
        self.num_centroids = num_centroids
        self.use_xyz = use_xyz
        self.mlp = SharedMLP(mlp_in + (3 if use_xyz else 0), mlp_channels, ndim=2, bn=True)
        self.sampler = FarthestPointSampler(num_centroids) if num_centroids > 0 else None
        if num_neighbours < 0:
            assert radius < 0.0
            self.grouper = None
        else:
            assert num_neighbours > 0 and radius > 0.0
            self.grouper = self._grouper_cls(radius, num_neighbours)

    def _global_group(self, xyz, feature):
        """num_centroids == 0: one group holding every point, centred at the origin."""
        assert self.grouper is None
        group_feature = feature.unsqueeze(2)
        if self.use_xyz:
            group_feature = torch.cat([xyz.unsqueeze(2), group_feature], dim=1)
        return xyz.new_zeros(xyz.size(0), 3, 1), group_feature

    def _sample(self, xyz):
        if self.num_centroids == -1:
            return None, xyz
        index = self.sampler(xyz)
        return index, _F.gather_points(xyz, index)

    def _reduce(self, x):
        return torch.max(x, 3)[0]

    def _mlp_reduce(self, group_feature):
        """SharedMLP then the reduction over the K neighbours (modules.py:244-245)."""
        if type(self)._reduce is _SetAbstraction._reduce:
            return self.mlp(group_feature, pool_max=True)   # max: fusable into the last BatchNorm + ReLU pass
        return self._reduce(self.mlp(group_feature))

    def forward(self, xyz, feature=None, geo=None):
        """``geo``: dict(new_xyz, nbr) of ``fused.sa_geometry`` when sampling and ball query ran ahead of time."""
        if self.num_centroids == 0:
            new_xyz, group_feature = self._global_group(xyz, feature)
        elif geo is not None and type(self.grouper) is QueryGrouper:
            new_xyz = geo["new_xyz"]
            group_feature, _ = self.grouper(new_xyz, xyz, feature, use_xyz=self.use_xyz, index=geo["nbr"])
        else:
            _, new_xyz = self._sample(xyz)
            group_feature, _ = self.grouper(new_xyz, xyz, feature, use_xyz=self.use_xyz)
        return new_xyz, self._mlp_reduce(group_feature)

    def init_weights(self, init_fn=None):
        self.mlp.init_weights(init_fn)

    def extra_repr(self):
        return "num_centroids={:d}, use_xyz={}".format(self.num_centroids, self.use_xyz)


class PointNetSAModule(_SetAbstraction):
    """PointNet++ set abstraction, max over the K neighbours (modules.py:176-252).

    xyz (B,3,N), feature (B,C,N) -> new_xyz (B,3,M), new_feature (B,C_out,M).  In eval mode
    under ``torch.no_grad()`` on a GPU the forward is served by the fused MI355X kernel chain
    (``fused.sa_forward``) which produces the same tensors without materialising the groups.
    """

    def __init__(self, in_channels, mlp_channels, num_centroids, radius, num_neighbours, use_xyz):
        super().__init__()
        self._build(in_channels, mlp_channels, num_centroids, radius, num_neighbours, use_xyz, in_channels)

    def forward(self, xyz, feature=None, geo=None):
        from .. import fused
        if (fused.usable(self, xyz) and self.num_centroids > 0 and self.grouper is not None
                and fused.supports_sa(self, feature)):
            return fused.sa_forward(self, xyz, feature, geo)
        return super().forward(xyz, feature, geo)


class PointNetSAAvgModule(_SetAbstraction):
    """Mean over neighbours instead of max (modules.py:255-331)."""

    def __init__(self, in_channels, mlp_channels, num_centroids, radius, num_neighbours, use_xyz):
        super().__init__()
        self._build(in_channels, mlp_channels, num_centroids, radius, num_neighbours, use_xyz, in_channels)

    def _reduce(self, x):
        return torch.mean(x, 3)


class EdgeSAModule(_SetAbstraction):
    """Edge-feature SA (modules.py:409-477): the MLP sees [xyz | f_j | f_j - f_i]."""

    _grouper_cls = EdgeQueryGrouper

    def __init__(self, in_channels, mlp_channels, num_centroids, radius, num_neighbours, use_xyz):
        super().__init__()
        self._build(in_channels, mlp_channels, num_centroids, radius, num_neighbours, use_xyz,
                    in_channels * 2 if num_centroids != 0 else in_channels)
        del self.in_channels  # the reference does not define it on this class

    def forward(self, xyz, feature=None):
        if self.num_centroids == 0:
            new_xyz, group_feature = self._global_group(xyz, feature)
        else:
            index, new_xyz = self._sample(xyz)
            if feature is None:
                centroid_feature = None
            elif index is None:
                centroid_feature = feature
            else:
                centroid_feature = _F.gather_points(feature, index)
            group_feature, _ = self.grouper(new_xyz, xyz, centroid_feature, feature, use_xyz=self.use_xyz)
        return new_xyz, self._reduce(self.mlp(group_feature))


class PointNetSAModuleMSG(nn.Module):
    """Multi-scale grouping SA: one (grouper, mlp) pair per radius, outputs concatenated
    (modules.py:334-406)."""

    def __init__(self, in_channels, mlp_channels_list, num_centroids, radius_list, num_neighbours_list, use_xyz):
        super().__init__()
        assert len(radius_list) == len(mlp_channels_list) == len(num_neighbours_list)
        self.in_channels = in_channels
        self.out_channels = sum(ch[-1] for ch in mlp_channels_list)
        self.num_centroids = num_centroids
        self.use_xyz = use_xyz
        self.mlp = nn.ModuleList()
        if num_centroids == -1:
            self.sampler = None
        else:
            assert num_centroids > 0
            self.sampler = FarthestPointSampler(num_centroids)
        self.grouper = nn.ModuleList()
        mlp_in = in_channels + (3 if use_xyz else 0)
        for channels, radius, k in zip(mlp_channels_list, radius_list, num_neighbours_list):
            self.mlp.append(SharedMLP(mlp_in, channels, ndim=2, bn=True))
            self.grouper.append(QueryGrouper(radius, k))

    def forward(self, xyz, feature=None):
        new_xyz = _F.gather_points(xyz, self.sampler(xyz)) if self.num_centroids > 0 else xyz
        outs = []
        for mlp, grouper in zip(self.mlp, self.grouper):
            group_feature, _ = grouper(new_xyz, xyz, feature, use_xyz=self.use_xyz)
            outs.append(torch.max(mlp(group_feature), 3)[0])
        return new_xyz, torch.cat(outs, dim=1)

    def init_weights(self, init_fn=None):
        for mlp in self.mlp:
            mlp.init_weights(init_fn)

    def extra_repr(self):
        return "num_centroids={:d}, use_xyz={}".format(self.num_centroids, self.use_xyz)


def _make_interpolator(num_neighbors, cls):
    if num_neighbors == 0:
        return None
    if num_neighbors == 3:
        return cls(num_neighbors)
    raise ValueError("Expected value 1 or 3, but {} given.".format(num_neighbors))


def _broadcast_global(dense_xyz, sparse_xyz, dense_feature, sparse_feature):
    """num_neighbors == 0: a single global feature is tiled over the dense points."""
    assert sparse_xyz.size(2) == 1 and sparse_feature.size(2) == 1
    return torch.cat([sparse_feature.expand(-1, -1, dense_xyz.size(2)), dense_feature], dim=1)


class PointnetFPModule(nn.Module):
    """PointNet++ feature propagation: 3-NN interpolate, concat skip features, SharedMLP
    (modules.py:480-512).  In eval mode under no_grad on a GPU the fused MI355X kernel chain
    serves the forward (``fused.fp_forward``)."""

    def __init__(self, in_channels, mlp_channels, num_neighbors):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, mlp_channels[-1]
        self.mlp = SharedMLP(in_channels, mlp_channels, ndim=1, bn=True)
        self.interpolator = _make_interpolator(num_neighbors, FeatureInterpolator)

    def forward(self, dense_xyz, sparse_xyz, dense_feature, sparse_feature, geo=None):
        if self.interpolator is None:
            return self.mlp(_broadcast_global(dense_xyz, sparse_xyz, dense_feature, sparse_feature))
        from .. import fused
        if fused.usable(self, dense_xyz) and fused.supports_fp(self, sparse_feature):
            return fused.fp_forward(self, dense_xyz, sparse_xyz, dense_feature, sparse_feature, geo)
        if geo is not None and type(self.interpolator) is FeatureInterpolator:
            return self.mlp(self.interpolator(dense_xyz, sparse_xyz, dense_feature, sparse_feature, geo))
        return self.mlp(self.interpolator(dense_xyz, sparse_xyz, dense_feature, sparse_feature))

    def init_weights(self, init_fn=None):
        self.mlp.init_weights(init_fn)


class EdgeFPModule(nn.Module):
    """Edge-feature FP (modules.py:515-549): 2-D SharedMLP over (N, K) then mean over K."""

    def __init__(self, in_channels, mlp_channels, num_neighbors):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, mlp_channels[-1]
        self.interpolator = _make_interpolator(num_neighbors, EdgeFeatureInterpolator)
        self.mlp = SharedMLP(in_channels, mlp_channels, ndim=1 if self.interpolator is None else 2, bn=True)

    def forward(self, dense_xyz, sparse_xyz, dense_feature, sparse_feature):
        if self.interpolator is None:
            return self.mlp(_broadcast_global(dense_xyz, sparse_xyz, dense_feature, sparse_feature))
        x = self.mlp(self.interpolator(dense_xyz, sparse_xyz, dense_feature, sparse_feature))
        return torch.mean(x, dim=-1)
