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


# Training on the GPU: evaluate a block's first (linear) layer before the gather / interpolation it commutes with
# (_SetAbstraction._premul_first_layer, PointnetFPModule._premul_first_layer).  Tests switch it off for A/B comparisons.
TRAIN_PREMUL = True


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


class _GroupMinus(torch.autograd.Function):
    """Y[b,c,m,k] = U[b,c,index[b,m,k]] - V[b,c,m] (the grouping of a pre-multiplied first layer, GPU training).  Backward:
    dU by the grouping's own scatter-add kernel, dV = -sum_k dY by one pass at HBM speed (torch's `neg` of the whole
    (B,C,M,K) gradient followed by a 4-D `sum` took 0.78 + 0.40 ms per iteration for levels 2-3)."""

    @staticmethod
    def forward(ctx, U, V, index):
        from .. import pn2_ext
        Y = pn2_ext.group_points_forward(U.contiguous(), index)
        Y -= V.unsqueeze(-1)
        ctx.save_for_backward(index)
        ctx.num_points = U.shape[2]
        return Y

    @staticmethod
    def backward(ctx, dY):
        from .. import pn2_ext, region_ops
        (index,) = ctx.saved_tensors
        dY = dY.contiguous()
        dU = pn2_ext.group_points_backward(dY, index, ctx.num_points) if ctx.needs_input_grad[0] else None
        dV = None
        if ctx.needs_input_grad[1]:
            K = dY.shape[-1]
            if 4 <= K <= 256 and K & (K - 1) == 0 and dY.data_ptr() % 16 == 0:
                dV = region_ops.rowsum_neg(dY, K)
            else:                         # neighbourhood sizes rowsum_neg_kernel has no instantiation for (e.g. K = 48)
                dV = -(dY.sum(-1))
        return dU, dV, None


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
        return index, _F.gather_sampled_points(xyz, index)

    def _reduce(self, x):
        return torch.max(x, 3)[0]

    def _premul_first_layer(self, xyz, feature, new_xyz, index):
        """Training on the GPU, wide inputs: the first shared-MLP layer is linear in [x_j - c | f_j], so
        W [x_j - c ; f_j] = (W [x_j ; f_j]) - (W_xyz c): evaluate it once per SOURCE point (U) and once per centroid (V)
        and group the products, instead of grouping the (3 + C)-wide input and multiplying every (centroid, neighbour)
        pair -- the (B, 3 + C, M, K) tensor, its concatenation copy and its gradient are never formed, and the GEMM
        (forward, input gradient, weight gradient) runs on N rows instead of M * K.  Same values up to fp32
        reassociation; returns the first convolution's output (B, C1, M, K) or None when this does not apply."""
        if not (TRAIN_PREMUL and self.training and xyz.is_cuda and feature is not None and self.use_xyz
                and type(self.grouper) is QueryGrouper and feature.shape[1] >= 32 and len(self.mlp) > 0):
            return None
        from .. import conv1x1_train
        block = self.mlp[0]
        conv = getattr(block, "conv", None)
        if conv is None or not conv1x1_train.supported(conv, feature.unsqueeze(-1)) or conv.weight.shape[1] != 3 + feature.shape[1]:
            return None
        W = conv.weight.view(conv.weight.shape[0], -1)                     # columns: [xyz | feature] (modules.py:52)
        src = torch.cat([xyz, feature], dim=1).contiguous()                # (B, 3 + C, N): source points, not groups
        U = conv1x1_train.gemm_conv(src, W)                                # (B, C1, N)
        V = conv1x1_train.gemm_conv(new_xyz.contiguous(), W[:, :3].contiguous())   # (B, C1, M)
        return _GroupMinus.apply(U, V, index)

    def _mlp_reduce(self, group_feature):
        """SharedMLP then the reduction over the K neighbours (modules.py:244-245)."""
        if type(self)._reduce is _SetAbstraction._reduce:
            return self.mlp(group_feature, pool_max=True)   # max: fusable into the last BatchNorm + ReLU pass
        return self._reduce(self.mlp(group_feature))

    def forward(self, xyz, feature=None, geo=None):
        """``geo``: dict(new_xyz, nbr) of ``fused.sa_geometry`` when sampling and ball query ran ahead of time."""
        if self.num_centroids == 0:
            new_xyz, group_feature = self._global_group(xyz, feature)
        else:
            index = None
            if geo is not None and type(self.grouper) is QueryGrouper:
                new_xyz, index = geo["new_xyz"], geo["nbr"]
            else:
                _, new_xyz = self._sample(xyz)
            if type(self.grouper) is QueryGrouper and type(self)._reduce is _SetAbstraction._reduce:
                if index is None and self.training and xyz.is_cuda:
                    with torch.no_grad():
                        index, _ = _F.ball_query(xyz, new_xyz, self.grouper.radius, self.grouper.num_neighbours)
                first = self._premul_first_layer(xyz, feature, new_xyz, index) if index is not None else None
                if first is not None:
                    return new_xyz, self.mlp(first, pool_max=True, first_affine_done=True)
            if index is not None:
                group_feature, _ = self.grouper(new_xyz, xyz, feature, use_xyz=self.use_xyz, index=index)
            else:
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
        new_xyz = _F.gather_sampled_points(xyz, self.sampler(xyz)) if self.num_centroids > 0 else xyz
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
        first = self._premul_first_layer(dense_xyz, sparse_xyz, dense_feature, sparse_feature, geo)
        if first is not None:
            return self.mlp(first, first_affine_done=True)
        if geo is not None and type(self.interpolator) is FeatureInterpolator:
            return self.mlp(self.interpolator(dense_xyz, sparse_xyz, dense_feature, sparse_feature, geo))
        return self.mlp(self.interpolator(dense_xyz, sparse_xyz, dense_feature, sparse_feature))

    def _premul_first_layer(self, dense_xyz, sparse_xyz, dense_feature, sparse_feature, geo):
        """Training on the GPU: the first layer is linear in [interpolated | skip] and the 3-NN interpolation is linear
        in the features, so multiply the SPARSE points (and the skip features by their own weight columns) and
        interpolate the products: the GEMM runs on N_sparse rows and the interpolated tensor is C1 wide instead of
        C_sparse.  Same values up to fp32 reassociation; None when it does not apply or would not pay."""
        if not (TRAIN_PREMUL and self.training and dense_xyz.is_cuda and type(self.interpolator) is FeatureInterpolator
                and len(self.mlp) > 0):
            return None
        from .. import conv1x1_train
        conv = getattr(self.mlp[0], "conv", None)
        Cs = sparse_feature.shape[1]
        Cd = 0 if dense_feature is None else dense_feature.shape[1]
        if (conv is None or not conv1x1_train.supported(conv, sparse_feature) or conv.weight.shape[1] != Cs + Cd
                or sparse_feature.shape[2] >= dense_xyz.shape[2] or conv.weight.shape[0] > Cs):
            return None
        W = conv.weight.view(conv.weight.shape[0], -1)                     # columns: [interpolated | skip] (modules.py:127)
        index, weight = _three_nn_weights(dense_xyz, sparse_xyz, self.interpolator.num_neighbors, self.interpolator._eps, geo)
        Ys = conv1x1_train.gemm_conv(sparse_feature.contiguous(), W[:, :Cs].contiguous())
        Y = _F.feature_interpolate(Ys, index, weight)
        if Cd:
            Y = Y + conv1x1_train.gemm_conv(dense_feature.contiguous(), W[:, Cs:].contiguous())
        return Y

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
