"""Network definitions (mirror of multi_model/utils/pointnet2.py:12-255).

``PointNet2Seg`` = the ScoreNet backbone (3 SA + 3 FP + shared-MLP head),
``PointNet2TwoStage`` = grasp-region head, ``PointNet2Refine`` = refine head.  Attribute names
match the reference so the 127 / 86 ``state_dict`` keys are identical.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .pn2_utils.modules import PointNetSAModule, PointnetFPModule
from .pn2_utils.nn import SharedMLP

# hyper-parameters hard-coded by the reference (pointnet2.py:40-46)
_SA_CENTROIDS = (5120, 1024, 256)
_SA_RADIUS = (0.02, 0.08, 0.32)
_SA_NEIGHBOURS = (64, 64, 64)
_SA_CHANNELS = ((128, 128, 256), (256, 256, 512), (512, 512, 1024))
_FP_CHANNELS = ((1024, 1024), (512, 512), (256, 256, 256))
_FP_NEIGHBOURS = (3, 3, 3)
_SEG_CHANNELS = (512, 256, 256, 128)


class PointNet2Seg(nn.Module):
    """points (B,6,N) -> (fp3 feature (B,256,N), score (B,N)).

    Note the first return value is the LAST FP output (256 ch), not the 128-ch head
    activation (pointnet2.py:121)."""

    _SA_MODULE = PointNetSAModule
    _FP_MODULE = PointnetFPModule

    def __init__(self, input_chann=3, k_score=1, k_obj=2, add_channel_flag=False, dropout_prob=0.5):
        super().__init__()
        self.k_score = k_score
        width = input_chann - 3
        skip = [width]
        self.sa_modules = nn.ModuleList()
        for m, r, k, channels in zip(_SA_CENTROIDS, _SA_RADIUS, _SA_NEIGHBOURS, _SA_CHANNELS):
            self.sa_modules.append(self._SA_MODULE(in_channels=width, mlp_channels=channels, num_centroids=m,
                                                   radius=r, num_neighbours=k, use_xyz=True))
            width = channels[-1]
            skip.append(width)
        self.fp_modules = nn.ModuleList()
        for level, (channels, k) in enumerate(zip(_FP_CHANNELS, _FP_NEIGHBOURS)):
            self.fp_modules.append(self._FP_MODULE(in_channels=width + skip[-2 - level], mlp_channels=channels,
                                                   num_neighbors=k))
            width = channels[-1]
        self.mlp = SharedMLP(width * 3 if add_channel_flag else width, _SEG_CHANNELS, ndim=1,
                             dropout_prob=dropout_prob)
        self.conv_score = nn.Conv1d(_SEG_CHANNELS[-1], self.k_score, 1)
        self.bn_score = nn.BatchNorm1d(self.k_score)
        self.sigmoid = nn.Sigmoid()

    def sample_level1(self, points):
        """Level-1 furthest point sampling only (5120 of N points): the longest latency chain of the
        forward, separable so a pipeline can run it two batches ahead.  Pass the result to ``plan``."""
        from . import fused
        xyz = points[:, :3, :]
        if not (fused.ENABLED and xyz.is_cuda):
            raise RuntimeError("PointNet2Seg.sample_level1 needs GPU tensors")
        with torch.no_grad():
            return fused.sa_sample(self.sa_modules[0], xyz)[0]

    def sample_levels(self, points, after_level=None):
        """Furthest point sampling of ALL set-abstraction levels (each level samples the previous level's centroids): the
        part of the geometry whose launches hold whole CUs, separable so that a pipeline can run it for many batches in
        one launch per level.  -> list of (B, M_level) index tensors; pass it to ``plan`` as ``level1_ctr``."""
        from . import fused
        xyz = points[:, :3, :]
        if not (fused.ENABLED and xyz.is_cuda):
            raise RuntimeError("PointNet2Seg.sample_levels needs GPU tensors")
        ctrs = []
        first_tie = None         # of the level above: levels 2+ sample that level's picks in pick order (fused.sa_sample)
        with torch.no_grad():
            for sa in self.sa_modules:
                ctr, first_tie = fused.sa_sample(sa, xyz, first_tie)
                ctrs.append(ctr)
                if after_level is not None:
                    after_level(len(ctrs) - 1)     # (a pipeline records an event here: level 1 is usable before levels 2-3 exist)
                xyz = torch.gather(xyz, 2, ctr[:, None, :].expand(xyz.shape[0], 3, ctr.shape[1]))
        return ctrs

    def plan(self, points, level1_ctr=None, on_level=None):
        """Geometry of a forward pass -- FPS / ball-query / 3-NN indices of every level (``level1_ctr``: the level-1
        sampling from ``sample_level1``, or the list of all levels' from ``sample_levels``, when already computed).  Depends on
        xyz only -- except the per-centre first-layer term ``V`` of levels 2-3 in eval mode, which is stored with the
        signature of the weights it was computed from and recomputed by ``fused.sa_features`` if they changed since -- so it
        can be computed ahead of (and concurrently with) the feature pass; hand the
        result to ``forward(points, plan=...)`` -- the fused inference path or the operator-granular training path.
        GPU only; always computed without autograd (the reference's geometry ops return no gradients)."""
        from . import fused
        xyz = points[:, :3, :]
        if not (fused.ENABLED and xyz.is_cuda):
            raise RuntimeError("PointNet2Seg.plan needs GPU tensors")
        with torch.no_grad():
            return self._plan(xyz, level1_ctr, on_level)

    def _plan(self, xyz, level1_ctr, on_level=None):
        """``on_level(kind, i, geo)`` (kind "sa" / "fp"; geo None before the level's launches, the level's dict after them):
        where a pipeline waits for a level's sampling and marks the level's geometry ready (geo["ready"], honoured by
        ``forward(plan=...)``), so that the first set-abstraction block need not wait for the deeper levels' geometry."""
        from . import fused
        levels, sa_geo = [xyz], []
        ctrs = list(level1_ctr) if isinstance(level1_ctr, (list, tuple)) else [level1_ctr]
        first_tie = None
        for i, sa in enumerate(self.sa_modules):
            if on_level is not None:
                on_level("sa", i, None)
            geo = fused.sa_geometry(sa, levels[-1], ctrs[i] if i < len(ctrs) else None, prefix_ok=first_tie)
            first_tie = geo.get("first_tie")    # None when this level's picks were handed in: the next level samples for real
            if on_level is not None:
                on_level("sa", i, geo)
            sa_geo.append(geo)
            levels.append(geo["new_xyz"])
        fp_geo, sparse = [], levels[-1]
        for level, fp in enumerate(self.fp_modules):
            dense = levels[-2 - level]
            fp_geo.append(fused.fp_geometry(fp, dense, sparse))
            if on_level is not None:
                on_level("fp", level, fp_geo[-1])
            sparse = dense
        return {"sa": sa_geo, "fp": fp_geo}

    def forward(self, points, add_channel1=None, add_channel2=None, plan=None):
        B, _, N = points.size()
        from . import fused
        xyz_stack, feat_stack = [points[:, :3, :]], [points[:, 3:6, :]]
        first_tie = None     # fused path: the sampling certificate of the level above (fused.sa_sample)
        for level, sa in enumerate(self.sa_modules):
            if plan is not None:
                _wait_ready(plan["sa"][level])
                xyz, feat = sa(xyz_stack[-1], feat_stack[-1], geo=plan["sa"][level])
            elif (type(sa) is PointNetSAModule and fused.usable(sa, xyz_stack[-1]) and sa.num_centroids > 0
                  and sa.grouper is not None and fused.supports_sa(sa, feat_stack[-1])):
                # the module's own fused forward, with its geometry made here so that levels 2+ know their input is the
                # level above's pick sequence (no sampling launch at all unless that run had a tie)
                geo = fused.sa_geometry(sa, xyz_stack[-1], prefix_ok=first_tie)
                first_tie = geo.get("first_tie")
                xyz, feat = sa(xyz_stack[-1], feat_stack[-1], geo=geo)
            else:
                first_tie = None
                xyz, feat = sa(xyz_stack[-1], feat_stack[-1])
            xyz_stack.append(xyz)
            feat_stack.append(feat)

        sparse_xyz, sparse_feature = xyz_stack[-1], feat_stack[-1]
        for level, fp in enumerate(self.fp_modules):
            dense_xyz = xyz_stack[-2 - level]
            if plan is not None:
                _wait_ready(plan["fp"][level])
            if (level == len(self.fp_modules) - 1 and add_channel1 is None and fused.usable(self, dense_xyz)
                    and fused.usable(fp, dense_xyz) and fused.supports_fp(fp, sparse_feature)):
                # last FP block + head as one chained kernel (csrc/rowchain.hip)
                geo = plan["fp"][level] if plan is not None else fused.fp_geometry(fp, dense_xyz, sparse_xyz)
                chained = fused.fp_head_forward(self, fp, dense_xyz, feat_stack[-2 - level], sparse_feature, geo)
                if chained is not None:
                    return chained
            if plan is not None:
                sparse_feature = fp(dense_xyz, sparse_xyz, feat_stack[-2 - level], sparse_feature,
                                    geo=plan["fp"][level])
            else:
                sparse_feature = fp(dense_xyz, sparse_xyz, feat_stack[-2 - level], sparse_feature)
            sparse_xyz = dense_xyz

        if add_channel1 is not None and add_channel2 is not None:
            extra = [c.view(B, 1, N).repeat(1, sparse_feature.shape[1], 1).float() for c in (add_channel1, add_channel2)]
            sparse_feature = torch.cat([sparse_feature] + extra, dim=1)

        if fused.usable(self, sparse_feature):
            return sparse_feature, fused.head_forward(self, sparse_feature)
        from . import bn_train, conv1x1_train
        x = self.mlp(sparse_feature)
        if self.training and conv1x1_train.small_co_ok(self.conv_score, x):
            x = conv1x1_train.conv1x1_small_co(self.conv_score, x)     # 128 -> k_score with bias: store-stream kernels, not MIOpen
        else:
            x = self.conv_score(x)
        if bn_train.supported(self.bn_score, x):
            # training on the GPU: the one-channel BatchNorm on this repo's passes (MIOpen's spatial kernels reduce the B x N
            # values of a single channel in one workgroup: 0.05 ms forward, 0.29 ms backward at 8 x 25 600)
            x = bn_train.bn_relu(self.bn_score, x, False)
        else:
            x = self.bn_score(x)
        score = self.sigmoid(x.transpose(2, 1).contiguous()).view(B, N)
        return sparse_feature, score


def _wait_ready(geo):
    """A plan level computed on another stream carries the event that marks it complete (pipeline.ForwardPipeline)."""
    ready = geo.get("ready")
    if ready is not None:
        torch.cuda.current_stream().wait_event(ready)


def _head_layer(conv, bn, x, relu):
    """``relu(bn(conv(x)))`` of the grasp heads, x (n, C, 1) with a data-dependent n (valid centres / grasps of the step).
    On the GPU MIOpen builds or looks up a kernel for every new (n, C, 1) -- ~10 ms per BatchNorm call and ~5 ms per
    convolution, 0.45 s per training iteration -- so there the 1x1 convolution is a plain GEMM over the rows and the
    batch norm uses torch's own kernels; on the CPU the reference's ops are kept."""
    if x.is_cuda and x.dim() == 3 and x.shape[2] == 1:
        y = F.linear(x.squeeze(2), conv.weight.squeeze(2), conv.bias)
        with torch.backends.cudnn.flags(enabled=False):
            y = bn(y)
        y = y.unsqueeze(2)
    else:
        y = bn(conv(x))
    return F.relu(y) if relu else y


class PointNet2TwoStage(nn.Module):
    """Grasp-region head (pointnet2.py:123-197): max-pool the grouped ScoreNet features of each
    centre, then a class branch (k_cls anchors) and a regression branch (k_reg values)."""

    def __init__(self, num_points, input_chann, k_cls, k_reg, k_reg_theta, add_channel_flag=False):
        super().__init__()
        self.num_points, self.k_reg, self.k_cls = num_points, k_reg, k_cls
        self.k_reg_no_anchor = self.k_reg // self.k_cls
        self.k_reg_theta = k_reg_theta

        self.conv = nn.Conv1d(256 * 3 if add_channel_flag else 256, 1024, 1)
        self.bn = nn.BatchNorm1d(1024)
        # registration order below fixes the state_dict key order (pointnet2.py:139-156)
        self.conv_cls2 = nn.Conv1d(1024, 256, 1)
        self.conv_cls3 = nn.Conv1d(256, 128, 1)
        self.linear_cls = nn.Linear(128, self.k_cls)  # never used in forward (reference quirk)
        self.conv_cls4 = nn.Conv1d(128, self.k_cls, 1)
        self.bn_cls2 = nn.BatchNorm1d(256)
        self.bn_cls3 = nn.BatchNorm1d(128)
        self.bn_cls4 = nn.BatchNorm1d(self.k_cls)
        self.conv_reg2 = nn.Conv1d(1024, 256, 1)
        self.conv_reg3 = nn.Conv1d(256, 128, 1)
        self.conv_reg4 = nn.Conv1d(128, self.k_reg, 1)
        self.bn_reg2 = nn.BatchNorm1d(256)
        self.bn_reg3 = nn.BatchNorm1d(128)
        self.bn_reg4 = nn.BatchNorm1d(self.k_reg)

        self.mp1 = nn.MaxPool1d(num_points)
        self.ap = nn.AdaptiveAvgPool1d(1)
        self.sigmod = nn.Sigmoid()

    def _branch(self, x, tag):
        for i in (2, 3):
            x = _head_layer(getattr(self, "conv_%s%d" % (tag, i)), getattr(self, "bn_%s%d" % (tag, i)), x, True)
        return _head_layer(getattr(self, "conv_%s4" % tag), getattr(self, "bn_%s4" % tag), x, False)

    def forward(self, xyz, feature, pooled=False, raw_reg=False):
        """xyz: grouped features (n, 256, num_points) -- or, with ``pooled=True``, the already
        max-pooled (n, 256, 1) tensor produced by the fused gather+max kernel.  ``raw_reg`` (the fused eval path only; whether
        it was honoured is left in ``self.reg_is_raw``): channels 7: of ``x_reg`` WITHOUT their sigmoid, for a caller whose
        decode kernel applies it."""
        mp_x = xyz if pooled else self.mp1(xyz)
        if feature is not None:
            mp_x = torch.cat((mp_x, feature.view(feature.shape[0], feature.shape[1], 1)), dim=1)
        from . import fused, heads_train
        self.reg_is_raw = False
        if self.training and mp_x.is_cuda:
            layers = heads_train._layers_twostage(self)
            if heads_train.supported(layers, mp_x):      # training on the GPU: the head as one autograd node (csrc/heads_train.hip)
                x_cls, x_reg = heads_train.twostage(self, mp_x)
                x_reg = x_reg.view(x_reg.shape[0], -1, self.k_reg_no_anchor)
                x_reg[:, :, 7:] = self.sigmod(x_reg[:, :, 7:])
                return x_cls, x_reg, mp_x
        if fused.usable(self, mp_x) and mp_x.shape[1] % 4 == 0:
            x_cls, x_reg = fused.twostage_forward(self, mp_x, raw_reg=raw_reg)
            self.reg_is_raw = bool(raw_reg)
            return x_cls, x_reg, mp_x
        x = _head_layer(self.conv, self.bn, mp_x, True)
        x_cls = self._branch(x, "cls")
        n, c, _ = x_cls.size()
        x_cls = x_cls.view(n, c)
        x_reg = self._branch(x, "reg").view(n, -1, self.k_reg_no_anchor)
        x_reg[:, :, 7:] = self.sigmod(x_reg[:, :, 7:])
        return x_cls, x_reg, mp_x


class PointNet2Refine(nn.Module):
    """Refine head (pointnet2.py:199-255): pooled gripper-box feature (+ region feature) ->
    2-way class and k_reg deltas."""

    def __init__(self, num_points=2500, input_chann=3, k_cls=2, k_reg=8):
        super().__init__()
        self.num_points, self.k_reg, self.k_cls = num_points, k_reg, k_cls
        self.conv_formal = nn.Conv1d(384, 1024, 1)
        self.bn_formal = nn.BatchNorm1d(1024)
        self.conv_formal_cls2 = nn.Conv1d(1024, 128, 1)
        self.conv_formal_cls3 = nn.Conv1d(128, self.k_cls, 1)
        self.bn_formal_cls2 = nn.BatchNorm1d(128)
        self.bn_formal_cls3 = nn.BatchNorm1d(self.k_cls)
        self.conv_formal_reg2 = nn.Conv1d(1024, 128, 1)
        self.conv_formal_reg3 = nn.Conv1d(128, self.k_reg, 1)
        self.bn_formal_reg2 = nn.BatchNorm1d(128)
        self.bn_formal_reg3 = nn.BatchNorm1d(self.k_reg)
        self.mp1 = nn.MaxPool1d(num_points)
        self.ap = nn.AdaptiveAvgPool1d(1)
        self.sigmoid = nn.Sigmoid()

    def _branch(self, x, tag):
        x = _head_layer(getattr(self, "conv_formal_%s2" % tag), getattr(self, "bn_formal_%s2" % tag), x, True)
        x = _head_layer(getattr(self, "conv_formal_%s3" % tag), getattr(self, "bn_formal_%s3" % tag), x, False)
        return x.view(x.shape[0], x.shape[1])

    def forward(self, gripper_feature, group_feature, pooled=False):
        x = gripper_feature if pooled else self.mp1(gripper_feature)
        if group_feature is not None:
            x = torch.cat((x, group_feature.view(group_feature.shape[0], group_feature.shape[1], 1)), dim=1)
        from . import fused, heads_train
        if self.training and x.is_cuda:
            layers = heads_train._layers_refine(self)
            if heads_train.supported(layers, x):
                return heads_train.refine(self, x)
        if fused.usable(self, x) and x.shape[1] % 4 == 0:
            return fused.refine_forward(self, x)
        x = _head_layer(self.conv_formal, self.bn_formal, x, True)
        return self._branch(x, "cls"), self._branch(x, "reg")
