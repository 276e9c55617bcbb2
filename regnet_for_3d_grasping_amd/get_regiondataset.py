"""Region grouping between ScoreNet and the grasp-region network (mirror of
dataset_utils/get_regiondataset.py:13-42, :279-295, :311-434).

``get_grasp_allobj`` keeps the reference's signature and return tuple.  What changes is where
the work happens: centre selection and the radius scan run on the GPU (FPS kernel, radius-group
kernel producing ascending candidate lists + counts); the host only draws the
``np.random.choice`` positions, in exactly the reference's call order (all scenes x centres of
the small-radius pass first, then the large-radius pass -- get_regiondataset.py:36-37), so a
seeded numpy RNG yields the same groups.  One device->host sync per pass instead of one per
centre.

Training labels (``_get_center_grasp``, get_regiondataset.py:45-134) need the dataset's grasp
pickles and are outside this round's scope: ``data_paths`` must be empty.
"""
import numpy as np
import torch

from . import np_random, region_ops
from .pn2_utils import function as _F


def get_grasp_allobj(pc, predict_score, params, data_paths, use_theta=True):
    """pc (B,N,6), predict_score (B,N), params = [center_num, score_thre, group_num, r_time_group,
    group_num_more, r_time_group_more, width, height, depth] ->
    (center_pc (B,Nc,6), center_pc_index (B,Nc), pc_group_index (B,Nc,G), pc_group (B,Nc,G,6),
     pc_group_more_index (B,Nc,Gm), pc_group_more (B,Nc,Gm,6), grasp_labels=None)."""
    (center_num, score_thre, group_num, r_time_group, group_num_more, r_time_group_more,
     width, height, depth) = params
    center_pc, center_pc_index = _select_score_center(pc, predict_score, center_num, score_thre)
    pc_group_index, pc_group = _get_group_pc(pc, center_pc, center_pc_index, group_num, width, height, depth,
                                             r_time_group)
    pc_group_more_index, pc_group_more = _get_group_pc(pc, center_pc, center_pc_index, group_num_more, width,
                                                       height, depth, r_time_group_more)
    if len(data_paths) > 0:
        raise NotImplementedError("grasp-label matching (_get_center_grasp) is not part of the forward hot path")
    return center_pc, center_pc_index, pc_group_index, pc_group, pc_group_more_index, pc_group_more, None


def _select_score_center(pc, pre_score, center_num, score_thre):
    """Pick ``center_num`` grasp centres per scene among points scoring > ``score_thre``
    (get_regiondataset.py:354-434): FPS over the positive subset when there are more than
    ``center_num`` positives (its first positive point is always centre 0); all positives padded
    with random repeats when 0 < P <= center_num; random points when P == 0.  The B == 1 and
    B > 1 branches of the reference implement the same rule and draw the same numpy variates.

    One host sync for the whole batch (the positive counts); the ascending positive indices of
    every scene come from one stable argsort instead of a ``torch.nonzero`` per scene."""
    B, N, C = pc.shape
    positive = pre_score.to(pc.device) > score_thre
    order = torch.argsort((~positive).to(torch.uint8), dim=1, stable=True)   # positives first, ascending
    counts = positive.sum(1).cpu().tolist()
    index = torch.empty((B, center_num), dtype=torch.int64, device=pc.device)
    for b in range(B):
        P = int(counts[b])
        map_index = order[b, :P]
        if P > center_num:
            sub_xyz = pc[b, map_index, :3].view(1, P, 3).transpose(2, 1)
            index[b] = map_index[_F.farthest_point_sample(sub_xyz, center_num).view(-1)]
        elif P > 0:
            extra = np.random.choice(P, center_num - P, replace=True)
            local = torch.cat([torch.arange(P), torch.from_numpy(np.asarray(extra, dtype=np.int64))])
            index[b] = map_index[local.to(pc.device)]
        else:
            picks = np.random.choice(N, center_num, replace=False)
            index[b] = torch.from_numpy(np.asarray(picks, dtype=np.int64)).to(pc.device)
    center_pc = torch.gather(pc, 1, index.unsqueeze(-1).expand(B, center_num, C))
    return center_pc, index


def group_radius(width, height, depth, r_time):
    """Radius as the reference's comparison sees it: the Python double ``max(w,h,d)*r_time``
    compared against float32 distances, i.e. rounded to float32 (get_regiondataset.py:291-294)."""
    return float(np.float32(max(width, height, depth) * r_time))


def _draw_positions(counts, group_num):
    """Host-side resampling of every (scene, centre) candidate list to exactly ``group_num``
    entries, consuming numpy's global RNG in the reference's order (get_regiondataset.py:331-337:
    scene-major, then centre; without replacement when ``n >= group_num`` else with).
    counts: (B,Nc) int array.  Returns positions (B,Nc,group_num) int64 into the ascending
    candidate lists; rows with no candidate are -1.  The draws run in native host code that is
    stream-compatible with numpy's legacy generator (np_random.choice_rows)."""
    return np_random.choice_rows(counts, group_num, 0)[0]


def _get_group_pc(pc, center_pc, center_pc_index, group_num, width, height, depth, r_time):
    """Radius grouping around every centre (get_regiondataset.py:311-352): candidates are the
    points with ``sqrt(dx^2+dy^2+dz^2) <= R`` (inclusive) in ascending index order, resampled to
    exactly ``group_num`` (without replacement when enough, else with).
    Returns pc_group_index (B,Nc,G) int64 and pc_group (B,Nc,G,C); empty groups stay -1."""
    B, N, C = pc.shape
    Nc = center_pc.shape[1]
    radius = group_radius(width, height, depth, r_time)
    cand, counts = region_ops.radius_candidates(pc, center_pc, radius)  # (B,Nc,cap) int32, (B,Nc) int32
    counts_np = counts.cpu().numpy()                                    # the one sync of this pass
    pos = torch.from_numpy(_draw_positions(counts_np, group_num)).to(pc.device)
    has_empty = bool((counts_np == 0).any())
    pc_group_index = torch.gather(cand, 2, pos.clamp(min=0)).long()
    if has_empty:  # candidate slots of an empty group were never written: do not gather through them
        empty = pos[:, :, :1] < 0
        pc_group_index = torch.where(empty, torch.zeros_like(pc_group_index), pc_group_index)
    flat = pc_group_index.view(B, Nc * group_num, 1).expand(B, Nc * group_num, C)
    pc_group = torch.gather(pc, 1, flat).view(B, Nc, group_num, C)
    if has_empty:
        pc_group_index = torch.where(empty, torch.full_like(pc_group_index, -1), pc_group_index)
        pc_group = torch.where(empty.unsqueeze(-1), torch.full_like(pc_group, -1.0), pc_group)
    return pc_group_index, pc_group
