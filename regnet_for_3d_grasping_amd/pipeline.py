"""The forward hot path end to end: ScoreNet -> region grouping -> grasp-region + refine
(the step of test.py:134-141 / train.py:358-367 without losses), as one callable used by
bench.py, smoke() and the tests."""
import contextlib
import io

import torch

from . import synthetic
from .get_regiondataset import get_grasp_allobj
from .gripper_region_network import GripperRegionNetwork
from .score_network import ScoreNetwork

# hyper-parameters hard-coded by the reference's train.py:70-90
CENTER_NUM, SCORE_THRE = 64, 0.5
GROUP_NUM, R_TIME_GROUP = 256, 0.1
GROUP_NUM_MORE, R_TIME_GROUP_MORE = 1024, 0.8
WIDTH, HEIGHT, DEPTH = 0.08, 0.01, 0.06
GRIPPER_NUM, GRASP_SCORE_THRESHOLD, REG_CHANNEL = 64, 0.5, 10

PARAMS = [CENTER_NUM, SCORE_THRE, GROUP_NUM, R_TIME_GROUP, GROUP_NUM_MORE, R_TIME_GROUP_MORE, WIDTH, HEIGHT, DEPTH]
GRIPPER_PARAMS = [WIDTH, HEIGHT, DEPTH]


def build_models(device, score_seed=7, region_seed=11):
    """ScoreNetwork + GripperRegionNetwork in eval mode with seeded (numpy) weights."""
    score_net = ScoreNetwork(training=True)
    score_net.load_state_dict(synthetic.seeded_state_dict(score_net, score_seed))
    region_net = GripperRegionNetwork(training=True, group_num=GROUP_NUM, gripper_num=GRIPPER_NUM,
                                      grasp_score_threshold=GRASP_SCORE_THRESHOLD, radius=DEPTH,
                                      reg_channel=REG_CHANNEL)
    region_net.load_state_dict(synthetic.seeded_state_dict(region_net, region_seed))
    return score_net.to(device).eval(), region_net.to(device).eval()


def forward_scenes(score_net, region_net, pc, with_region=True):
    """pc (B,N,6) -> dict(all_feature, score, centres, next_grasp, select_grasp_class, ...)."""
    with torch.no_grad():
        all_feature, score, _ = score_net(pc)
        out = {"all_feature": all_feature, "score": score}
        if not with_region:
            return out
        (center_pc, center_idx, g_idx, g, gm_idx, gm, _) = get_grasp_allobj(pc, score, PARAMS, [])
        with contextlib.redirect_stdout(io.StringIO()):
            res = region_net(g, gm, g_idx, gm_idx, center_pc, center_idx, pc, all_feature, GRIPPER_PARAMS, None, [])
    out.update(center_pc_index=center_idx, pc_group_index=g_idx, pc_group_more_index=gm_idx, next_grasp=res[0],
               select_grasp_class=res[6], select_grasp_score=res[7], final_mask=res[11])
    return out
