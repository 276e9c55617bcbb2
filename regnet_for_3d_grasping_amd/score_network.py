"""ScoreNet wrapper (mirror of multi_model/score_network.py:9-53)."""
from torch import nn

from .pointnet2 import PointNet2Seg


class ScoreNetwork(nn.Module):
    """pc (B,N,6) -> (all_feature (B,N,256), output_score (B,N), loss or None)."""

    def __init__(self, training=True, k_obj=2):
        super().__init__()
        self.is_training = training
        self.k_obj = k_obj
        self.extrat_featurePN2 = PointNet2Seg(input_chann=6, k_score=1, k_obj=self.k_obj)
        self.criterion_cls = nn.NLLLoss(reduction="mean")
        self.criterion_reg = nn.MSELoss(reduction="mean")

    def compute_loss(self, pscore, tscore):
        """MSE between predicted and target per-point score (score_network.py:18-29)."""
        return self.criterion_reg(pscore, tscore.float())

    def sample_level1(self, pc):
        """Level-1 FPS indices for ``pc``; see PointNet2Seg.sample_level1."""
        return self.extrat_featurePN2.sample_level1(pc[:, :, :6].permute(0, 2, 1))

    def sample_levels(self, pc, after_level=None):
        """FPS indices of every set-abstraction level for ``pc``; see PointNet2Seg.sample_levels."""
        return self.extrat_featurePN2.sample_levels(pc[:, :, :6].permute(0, 2, 1), after_level)

    def plan(self, pc, level1_ctr=None, on_level=None):
        """Geometry plan (sampling / grouping / 3-NN indices) for ``pc``; see PointNet2Seg.plan."""
        return self.extrat_featurePN2.plan(pc[:, :, :6].permute(0, 2, 1), level1_ctr, on_level)

    def forward(self, pc, pc_score=None, pc_label=None, plan=None):
        points = pc[:, :, :6].permute(0, 2, 1)
        if plan is not None:
            feature, output_score = self.extrat_featurePN2(points, plan=plan)
        else:
            feature, output_score = self.extrat_featurePN2(points)
        all_feature = feature.transpose(2, 1)
        loss = None
        if self.is_training and pc_score is not None:
            loss = self.compute_loss(output_score, pc_score)
        return all_feature, output_score, loss
