"""Full-size golden fixtures from the REFERENCE's own Python graph (/root/reference, read-only, never copied) over
the CPU oracle, authoring container only:

    python tests/golden/make_golden_fullsize.py        # writes tests/golden/s7_*.npz / s7_meta.json

tests/golden/make_golden.py pins the path at B=2 x N=6 144; this script pins it at the sizes BASELINE.json's configs
are quoted on, so that the 25 600- and 51 200-point checks compare the HIP path with REFERENCE output and not only
with this repo's own oracle-backed mirror:

  S7a  ScoreNet forward, B=4 x 25 600 (configs[1]; scene 0 alone is the B=1 case of configs[0]; eval-mode batches are
       scene-independent): SHA-256 of every FPS / ball-query / 3-NN index tensor, the full score tensor, a strided
       sample of the 256-channel feature
  S7b  ScoreNet forward, B=1 x 51 200 (the cloud size of configs[4]): same contents
  S7c  region grouping + grasp-region / refine forward at 25 600 points (configs[2]), B=2, teacher-forced from S7a's
       REFERENCE scores of scenes 0-1 and a seeded pseudo feature map (as S2/S3 of make_golden.py): centre indices,
       SHA-256 of both group index tensors, numpy RNG state, grasps, masks, refine outputs

Inputs are regenerated from seeds by regnet_for_3d_grasping_amd.synthetic; fixtures hold expected OUTPUTS only.
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402
import make_golden as mg  # noqa: E402

sys.path.insert(0, _ref_shims.REPO_ROOT)
from regnet_for_3d_grasping_amd import synthetic  # noqa: E402

CFG = dict(a=dict(B=4, N=25600, scene_seed=2000), b=dict(B=1, N=51200, scene_seed=2100),
           score_weights_seed=7, region_weights_seed=11, c=dict(B=2, feature_seed=43, np_seed=777),
           params=mg.CFG["params"], gripper_params=mg.CFG["gripper_params"], gripper_num=64,
           grasp_score_threshold=0.5, reg_channel=10, feature_stride_a=512, feature_stride_b=1024)


def scorenet_case(sn, ext_log, case, bn=None):
    B, N = case["B"], case["N"]
    pc = synthetic.make_batch(case["scene_seed"], B, N)
    net = sn.ScoreNetwork(training=True)
    net.load_state_dict(synthetic.seeded_state_dict(net, CFG["score_weights_seed"]))
    net.eval()
    if bn is None:
        mean, var = mg.calibrate_score_head(net, pc[:1])
        bn = {"running_mean": mean, "running_var": var, "weight": 2.0, "bias": 0.0}
    else:
        seg = net.extrat_featurePN2
        seg.bn_score.running_mean.fill_(bn["running_mean"])
        seg.bn_score.running_var.fill_(bn["running_var"])
        seg.bn_score.weight.data.fill_(bn["weight"])
        seg.bn_score.bias.data.fill_(bn["bias"])
    ext_log.clear()
    t0 = time.time()
    with torch.no_grad():
        all_feature, score, loss = net(pc)
    assert loss is None
    # per-scene digests (eval-mode scenes are independent), so a B=1 run of scene 0 can be checked against them too
    ops = [{"op": name, "shape": list(outs[0].shape), "index_sha256": [mg.sha(outs[0][b]) for b in range(B)],
            "aux_sha256": [mg.sha(outs[1][b]) for b in range(B)] if len(outs) > 1 else None}
           for name, outs in ext_log]
    print("B=%d N=%d: %.1f s, positives %s, score range %.4f..%.4f" % (
        B, N, time.time() - t0, [int(v) for v in (score > 0.5).sum(1)], float(score.min()), float(score.max())))
    return pc, all_feature, score, ops, bn


def main():
    sn, grn, grd = _ref_shims.import_reference()
    from oracle import pn2_ext_oracle as ext
    log = []
    for name in ("farthest_point_sample", "ball_query", "point_search"):
        orig = getattr(ext, name)

        def wrapped(*a, _orig=orig, _name=name):
            out = _orig(*a)
            outs = out if isinstance(out, (list, tuple)) else [out]
            log.append((_name, [o for o in outs]))
            return out
        setattr(ext, name, wrapped)

    meta = {"cfg": CFG, "torch": torch.__version__}
    # ---- S7a -------------------------------------------------------------------------------------------------
    pc_a, feat_a, score_a, ops_a, bn = scorenet_case(sn, log, CFG["a"])
    meta["bn_score"] = bn
    meta["s7a_ops"] = ops_a
    meta["s7a_positive"] = [int(v) for v in (score_a > 0.5).sum(1)]
    s7a = {"score": score_a.numpy(), "feature_sample": feat_a[:, ::CFG["feature_stride_a"], :].contiguous().numpy()}
    # ---- S7b (same head calibration: the weights are the same network) ------------------------------------------
    pc_b, feat_b, score_b, ops_b, _ = scorenet_case(sn, log, CFG["b"], bn)
    meta["s7b_ops"] = ops_b
    meta["s7b_positive"] = [int(v) for v in (score_b > 0.5).sum(1)]
    s7b = {"score": score_b.numpy(), "feature_sample": feat_b[:, ::CFG["feature_stride_b"], :].contiguous().numpy()}
    # ---- S7c: grouping + heads at 25 600 points, teacher-forced ---------------------------------------------------
    Bc, N = CFG["c"]["B"], CFG["a"]["N"]
    pc, score = pc_a[:Bc].contiguous(), score_a[:Bc].contiguous()
    np.random.seed(CFG["c"]["np_seed"])
    (center_pc, center_pc_index, pc_group_index, pc_group, pc_group_more_index, pc_group_more,
     labels) = grd.get_grasp_allobj(pc, score, CFG["params"], [])
    assert labels is None
    meta["s7c_group"] = {"center_pc_sha256": mg.sha(center_pc.float()),
                         "pc_group_index_sha256": mg.sha(pc_group_index.long()),
                         "pc_group_more_index_sha256": mg.sha(pc_group_more_index.long()),
                         "np_state_after": int(np.random.randint(0, 2 ** 31 - 1))}
    feat = mg.pseudo_feature(CFG["c"]["feature_seed"], Bc, N)
    rnet = grn.GripperRegionNetwork(training=True, group_num=CFG["params"][2], gripper_num=CFG["gripper_num"],
                                    grasp_score_threshold=CFG["grasp_score_threshold"],
                                    radius=CFG["gripper_params"][2], reg_channel=CFG["reg_channel"])
    rnet.load_state_dict(synthetic.seeded_state_dict(rnet, CFG["region_weights_seed"]))
    rnet.eval()
    np.random.seed(CFG["c"]["np_seed"] + 1)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        out = rnet(pc_group, pc_group_more, pc_group_index, pc_group_more_index, center_pc, center_pc_index, pc,
                   feat, CFG["gripper_params"], None, [])
    (next_grasp, keep2, true_mask, _, _, _, sel_class, sel_score, sel_class_s2, keep3, keep3s, final_mask,
     final_mask_sthre, _, _, _) = out
    empty = np.zeros((0, 10), np.float32)
    s7c = {"center_pc_index": center_pc_index.long().numpy(), "next_grasp": next_grasp.numpy(),
           "true_mask": true_mask.numpy(),
           "select_grasp_class": sel_class.numpy() if sel_class is not None else empty,
           "select_grasp_score": sel_score.numpy() if sel_score is not None else empty,
           "final_mask": final_mask.numpy() if final_mask is not None else np.zeros((0,), np.int64)}
    meta["s7c"] = {"keep2": [int(k) for k in keep2], "keep3": [int(k) for k in keep3],
                   "refine_ran": sel_class is not None, "np_state_after": int(np.random.randint(0, 2 ** 31 - 1))}
    print("S7c centres[0,:8]", s7c["center_pc_index"][0, :8], "class-1 grasps:", len(s7c["select_grasp_class"]))

    np.savez_compressed(os.path.join(HERE, "s7a_scorenet_25600.npz"), **s7a)
    np.savez_compressed(os.path.join(HERE, "s7b_scorenet_51200.npz"), **s7b)
    np.savez_compressed(os.path.join(HERE, "s7c_region_25600.npz"), **s7c)
    with open(os.path.join(HERE, "s7_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for fn in ("s7a_scorenet_25600.npz", "s7b_scorenet_51200.npz", "s7c_region_25600.npz", "s7_meta.json"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


if __name__ == "__main__":
    main()
