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


class ForwardPipeline:
    """Software pipeline over a stream of scene batches (inference).

    Scenes are independent, and inside one batch the sampling / grouping geometry depends on xyz
    only.  Furthest point sampling is a latency-bound chain of ~6400 dependent rounds that keeps
    one CU per scene busy, while the shared-MLP contraction wants the other ~250 CUs; region
    grouping needs the host for numpy's RNG.  So three stages run concurrently on three HIP
    streams, each on a different batch:

        s_geo : geometry(batch i+1)   FPS x3, ball query x3, 3-NN x3            (~8 CUs)
        s_mlp : features(batch i)     gather / MFMA shared-MLP / pool / head    (matrix cores)
        s_reg : region(batch i-1)     radius grouping, host RNG draws, GRN + refine heads

    Results are identical to running ``forward_scenes`` batch by batch (same kernels, same numpy
    RNG call order: region stages execute in batch order on the host thread).
    """

    def __init__(self, score_net, region_net, with_region=True):
        self.score_net, self.region_net, self.with_region = score_net, region_net, with_region
        dev = next(score_net.parameters()).device
        self.device = dev
        # Priorities: the region stage is a chain of small kernels separated by host syncs (numpy
        # RNG draws) and FPS is a single-CU latency chain -- neither may queue behind the big MLP
        # launches, so both get high-priority HW queues and the MFMA stream the default one.
        self.s_geo = torch.cuda.Stream(dev, priority=-1)
        self.s_mlp = torch.cuda.Stream(dev, priority=0)
        self.s_reg = torch.cuda.Stream(dev, priority=-1)

    # -- stages -------------------------------------------------------------------------------
    def _geometry(self, pc):
        from . import fused
        with torch.cuda.stream(self.s_geo), torch.no_grad():
            plan = self.score_net.plan(pc)
            done = torch.cuda.Event()
            done.record(self.s_geo)
        for t in fused.plan_tensors(plan):
            t.record_stream(self.s_mlp)
        return {"pc": pc, "plan": plan, "geo_done": done}

    def _features(self, item):
        with torch.cuda.stream(self.s_mlp), torch.no_grad():
            self.s_mlp.wait_event(item["geo_done"])
            all_feature, score, _ = self.score_net(item["pc"], plan=item["plan"])
            done = torch.cuda.Event()
            done.record(self.s_mlp)
        all_feature.record_stream(self.s_reg)
        score.record_stream(self.s_reg)
        item.update(all_feature=all_feature, score=score, mlp_done=done)
        item.pop("plan")
        return item

    def _region(self, item):
        pc, all_feature, score = item["pc"], item["all_feature"], item["score"]
        out = {"all_feature": all_feature, "score": score}
        with torch.cuda.stream(self.s_reg), torch.no_grad():
            self.s_reg.wait_event(item["mlp_done"])
            if self.with_region:
                (center_pc, center_idx, g_idx, g, gm_idx, gm, _) = get_grasp_allobj(pc, score, PARAMS, [])
                with contextlib.redirect_stdout(io.StringIO()):
                    res = self.region_net(g, gm, g_idx, gm_idx, center_pc, center_idx, pc, all_feature,
                                          GRIPPER_PARAMS, None, [])
                out.update(center_pc_index=center_idx, pc_group_index=g_idx, pc_group_more_index=gm_idx,
                           next_grasp=res[0], select_grasp_class=res[6], select_grasp_score=res[7],
                           final_mask=res[11])
            done = torch.cuda.Event()
            done.record(self.s_reg)
        out["done"] = done
        return out

    # -- driver -------------------------------------------------------------------------------
    def run(self, batches):
        """batches: iterable of (B,N,6) GPU tensors (already resident).  Yields one result dict per
        batch, in order.  The caller must ``result['done'].synchronize()`` (or synchronise the
        device) before reading results on another stream."""
        cur = torch.cuda.current_stream(self.device)
        for s in (self.s_geo, self.s_mlp, self.s_reg):
            s.wait_stream(cur)
        stage1 = stage2 = None
        it = iter(batches)
        exhausted = False
        while True:
            pc = None
            if not exhausted:
                try:
                    pc = next(it)
                except StopIteration:
                    exhausted = True
            if pc is None and stage1 is None and stage2 is None:
                break
            new1 = self._geometry(pc) if pc is not None else None          # async, s_geo
            new2 = self._features(stage1) if stage1 is not None else None  # async, s_mlp (after geo event)
            if stage2 is not None:
                yield self._region(stage2)                                 # host-heavy, s_reg
            stage1, stage2 = new1, new2
        for s in (self.s_geo, self.s_mlp, self.s_reg):
            cur.wait_stream(s)
