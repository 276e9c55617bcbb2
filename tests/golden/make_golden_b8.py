"""configs[2] AT ITS OWN BATCH SIZE from the REFERENCE's Python graph (/root/reference, read-only, never copied) over
the CPU oracle, authoring container only:

    python tests/golden/make_golden_b8.py        # writes tests/golden/s8_b8_25600.npz / s8_meta.json

S8  ScoreNet + region grouping + GraspRegionNet + RefineNet forward on 8 x 25 600 points -- the reference graph END TO
    END (the reference's own scores select the centres, its own 256-channel feature map feeds the heads), for the three
    batches tests/test_gpu_pipeline_b8.py sends through ``ForwardPipeline`` (the bench path): the 8 scenes in three
    different orders.  Eval-mode scenes are independent, so ScoreNet runs once over the 8 scenes; the region stage runs
    per batch with numpy re-seeded per batch (``np_seed + batch index``: a crop count that flipped on fp32 noise in one
    batch must not shift the stream of the next).

    Stored: per-scene SHA-256 of every FPS / ball-query / 3-NN index tensor, the full score tensor, a strided feature
    sample; per batch: centre indices, SHA-256 of both group index tensors, the numpy stream position after the grouping
    and after the heads, ``next_grasp``, ``true_mask``, ``keep2``, class-1 grasps.

The weights, scenes 0-3 and the score-head calibration are S7a's (make_golden_fullsize.py), so s7a's fixtures and these
agree on the scenes they share (asserted below).  Inputs are regenerated from seeds; fixtures hold expected OUTPUTS only.
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
import make_golden_fullsize as mf  # noqa: E402

sys.path.insert(0, _ref_shims.REPO_ROOT)
from regnet_for_3d_grasping_amd import synthetic  # noqa: E402

CFG = dict(B=8, N=25600, scene_seed=2000, np_seed=880, feature_stride=1024,
           orders=[[0, 1, 2, 3, 4, 5, 6, 7], [7, 6, 5, 4, 3, 2, 1, 0], [3, 0, 5, 2, 7, 4, 1, 6]])


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

    with open(os.path.join(HERE, "s7_meta.json")) as f:
        m7 = json.load(f)
    full = m7["cfg"]
    assert full["a"]["scene_seed"] == CFG["scene_seed"] and full["a"]["N"] == CFG["N"]
    mf.CFG["score_weights_seed"] = full["score_weights_seed"]
    case = dict(B=CFG["B"], N=CFG["N"], scene_seed=CFG["scene_seed"])
    pc, feat, score, ops, _ = mf.scorenet_case(sn, log, case, m7["bn_score"])
    # scenes 0-3 are S7a's: same reference output
    s7a = dict(np.load(os.path.join(HERE, "s7a_scorenet_25600.npz")))
    # (same reference output up to the fp32 re-association of torch's CPU convolutions at another batch size)
    d7 = float(np.abs(score[:4].numpy() - s7a["score"]).max())
    print("S8 vs S7a scores, scenes 0-3: max abs diff %.2e" % d7)
    assert d7 < 5e-6, "S8 scenes 0-3 differ from S7a"
    for a, b in zip(ops, m7["s7a_ops"]):
        assert a["op"] == b["op"] and a["index_sha256"][:4] == b["index_sha256"]

    meta = {"cfg": CFG, "torch": torch.__version__, "ops": ops,
            "positive": [int(v) for v in (score > 0.5).sum(1)], "batches": []}
    out_npz = {"score": score.numpy(),
               "feature_sample": feat[:, ::CFG["feature_stride"], :].contiguous().numpy()}

    rnet = grn.GripperRegionNetwork(training=True, group_num=full["params"][2], gripper_num=full["gripper_num"],
                                    grasp_score_threshold=full["grasp_score_threshold"],
                                    radius=full["gripper_params"][2], reg_channel=full["reg_channel"])
    rnet.load_state_dict(synthetic.seeded_state_dict(rnet, full["region_weights_seed"]))
    rnet.eval()
    empty = np.zeros((0, 10), np.float32)
    for bi, order in enumerate(CFG["orders"]):
        t0 = time.time()
        idx = torch.tensor(order)
        pcb, scb, ftb = pc[idx].contiguous(), score[idx].contiguous(), feat[idx].contiguous()
        np.random.seed(CFG["np_seed"] + bi)
        (center_pc, center_pc_index, g_idx, g, gm_idx, gm, labels) = grd.get_grasp_allobj(pcb, scb, full["params"], [])
        assert labels is None
        rec = {"center_pc_sha256": mg.sha(center_pc.float()), "pc_group_index_sha256": mg.sha(g_idx.long()),
               "pc_group_more_index_sha256": mg.sha(gm_idx.long()),
               "np_state_after_grouping": int(np.random.get_state()[2]),
               "np_word_after_grouping": int(np.random.get_state()[1][np.random.get_state()[2] % 624])}
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            out = rnet(g, gm, g_idx, gm_idx, center_pc, center_pc_index, pcb, ftb, full["gripper_params"], None, [])
        (next_grasp, keep2, true_mask, _, _, _, sel_class, sel_score, _, keep3, _, final_mask, _, _, _, _) = out
        rec.update(keep2=[int(k) for k in keep2], keep3=[int(k) for k in keep3], refine_ran=sel_class is not None,
                   np_draw_after=int(np.random.randint(0, 2 ** 31 - 1)))
        meta["batches"].append(rec)
        out_npz["b%d_center_pc_index" % bi] = center_pc_index.long().numpy()
        out_npz["b%d_next_grasp" % bi] = next_grasp.numpy()
        out_npz["b%d_true_mask" % bi] = true_mask.numpy()
        out_npz["b%d_select_grasp_class" % bi] = sel_class.numpy() if sel_class is not None else empty
        out_npz["b%d_final_mask" % bi] = final_mask.numpy() if final_mask is not None else np.zeros((0,), np.int64)
        print("batch %d: %.1f s, keep2 %s, class-1 grasps %d" % (bi, time.time() - t0, rec["keep2"],
                                                                 len(out_npz["b%d_select_grasp_class" % bi])))

    np.savez_compressed(os.path.join(HERE, "s8_b8_25600.npz"), **out_npz)
    with open(os.path.join(HERE, "s8_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for fn in ("s8_b8_25600.npz", "s8_meta.json"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


if __name__ == "__main__":
    main()
