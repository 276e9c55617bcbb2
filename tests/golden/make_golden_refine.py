"""configs[2] WITH THE REFINE STAGE RUNNING, from the REFERENCE's Python graph (/root/reference, read-only, never copied)
over the CPU oracle, authoring container only:

    python tests/golden/make_golden_refine.py        # writes tests/golden/s9_refine_b8.npz / s9_meta.json

S9  ScoreNet + region grouping + GraspRegionNet + RefineNet forward on 8 x 25 600 points as S8 (make_golden_b8.py: same
    scenes, ScoreNet weights and score-head calibration), but with the region head CALIBRATED
    (``synthetic.calibrate_region_head`` applied to the REFERENCE's module on batch 0): with purely seeded weights every
    S7 / S8 batch has fewer than two valid crops, so the reference's ``if len(gripper_mask) >= 2``
    (gripper_region_network.py:333) skips the whole refine network and "empty equals empty" is all those fixtures can
    check.  Here ~85 % of the 512 crops per batch are valid and about half of them come out as class 1.

    Stored per batch (two scene orders): centre indices, SHA-256 of both group index tensors, numpy stream position after
    the grouping; ``next_grasp`` (512 x 10); the crop stage: valid crop ids, candidate count of every valid crop (the
    first argument of the reference's ``np.random.choice`` calls, :535-537), SHA-256 + full array of
    ``gripper_pc_index_inall``; the refine stage: ``select_grasp_class / score / class_stage2``, ``final_mask``,
    ``final_mask_sthre``, keep counts; one numpy draw after the heads (stream position).
    The sixteen calibrated BatchNorm tensors are stored in s9_meta.json and loaded by the tests (never re-derived).

Inputs are regenerated from seeds; fixtures hold expected OUTPUTS only.
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

CFG = dict(B=8, N=25600, scene_seed=2000, np_seed=990, calibration_np_seed=989,
           orders=[[0, 1, 2, 3, 4, 5, 6, 7], [5, 2, 7, 0, 3, 6, 1, 4]])


def main():
    sn, grn, grd = _ref_shims.import_reference()
    with open(os.path.join(HERE, "s7_meta.json")) as f:
        m7 = json.load(f)
    full = m7["cfg"]
    mf.CFG["score_weights_seed"] = full["score_weights_seed"]
    case = dict(B=CFG["B"], N=CFG["N"], scene_seed=CFG["scene_seed"])
    pc, feat, score, _, _ = mf.scorenet_case(sn, [], case, m7["bn_score"])
    s8 = np.load(os.path.join(HERE, "s8_b8_25600.npz"))
    d8 = float(np.abs(score.numpy() - s8["score"]).max())
    print("S9 vs S8 scores: max abs diff %.2e" % d8)
    assert d8 == 0.0, "the ScoreNet leg must reproduce S8 exactly (same graph, same batch)"

    rnet = grn.GripperRegionNetwork(training=True, group_num=full["params"][2], gripper_num=full["gripper_num"],
                                    grasp_score_threshold=full["grasp_score_threshold"],
                                    radius=full["gripper_params"][2], reg_channel=full["reg_channel"])
    rnet.load_state_dict(synthetic.seeded_state_dict(rnet, full["region_weights_seed"]))
    rnet.eval()

    # spies on the reference's crop stage: its outputs, and the candidate count behind every np.random.choice
    crop = {}
    orig_transform = grn.get_gripper_region_transform

    def spy_transform(*a, **k):
        counts = []
        orig_choice = np.random.choice

        def choice(n, *ca, **ck):
            counts.append(int(n))
            return orig_choice(n, *ca, **ck)
        np.random.choice = choice
        try:
            out = orig_transform(*a, **k)
        finally:
            np.random.choice = orig_choice
        crop.update(index_inall=out[2].long().numpy().copy(), valid=out[3].long().numpy().copy(),
                    counts=np.asarray(counts, np.int64))
        return out
    grn.get_gripper_region_transform = spy_transform

    def region_forward(order, np_seed):
        idx = torch.tensor(order)
        pcb, scb, ftb = pc[idx].contiguous(), score[idx].contiguous(), feat[idx].contiguous()
        np.random.seed(np_seed)
        grouped = grd.get_grasp_allobj(pcb, scb, full["params"], [])
        (center_pc, center_pc_index, g_idx, g, gm_idx, gm, labels) = grouped
        assert labels is None
        after_grouping = np.random.get_state()
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            out = rnet(g, gm, g_idx, gm_idx, center_pc, center_pc_index, pcb, ftb, full["gripper_params"], None, [])
        return grouped, after_grouping, out

    constants = synthetic.calibrate_region_head(rnet, lambda: region_forward(CFG["orders"][0], CFG["calibration_np_seed"]))
    meta = {"cfg": CFG, "torch": torch.__version__,
            "region_calibration": {k: v.double().tolist() for k, v in constants.items()}, "batches": []}
    # float32 -> double -> JSON -> float32 is exact
    for k, v in constants.items():
        assert torch.equal(torch.tensor(meta["region_calibration"][k], dtype=torch.float64).float(), v.float())
    out_npz = {}
    for bi, order in enumerate(CFG["orders"]):
        t0 = time.time()
        grouped, after_grouping, out = region_forward(order, CFG["np_seed"] + bi)
        (center_pc, center_pc_index, g_idx, g, gm_idx, gm, _) = grouped
        (next_grasp, keep2, true_mask, _, _, _, sel_class, sel_score, sel_class_s2, keep3, keep3s, final_mask,
         final_mask_sthre, _, _, _) = out
        assert sel_class is not None, "the refine stage did not run"
        rec = {"center_pc_sha256": mg.sha(center_pc.float()), "pc_group_index_sha256": mg.sha(g_idx.long()),
               "pc_group_more_index_sha256": mg.sha(gm_idx.long()),
               "np_state_after_grouping": int(after_grouping[2]),
               "keep2": [int(k) for k in keep2], "keep3": [int(k) for k in keep3],
               "keep3_score": [int(k) for k in keep3s], "refine_ran": True,
               "valid_crops": int(len(crop["valid"])), "crop_index_inall_sha256": mg.sha(torch.from_numpy(crop["index_inall"])),
               "np_draw_after": int(np.random.randint(0, 2 ** 31 - 1))}
        meta["batches"].append(rec)
        p = "b%d_" % bi
        out_npz[p + "center_pc_index"] = center_pc_index.long().numpy()
        out_npz[p + "next_grasp"] = next_grasp.numpy()
        out_npz[p + "true_mask"] = true_mask.numpy()
        out_npz[p + "crop_valid"] = crop["valid"]
        out_npz[p + "crop_counts"] = crop["counts"]
        out_npz[p + "crop_index_inall"] = crop["index_inall"].astype(np.int32)
        out_npz[p + "select_grasp_class"] = sel_class.numpy()
        out_npz[p + "select_grasp_score"] = sel_score.numpy()
        out_npz[p + "select_grasp_class_stage2"] = sel_class_s2.numpy()
        out_npz[p + "final_mask"] = final_mask.numpy()
        out_npz[p + "final_mask_sthre"] = final_mask_sthre.numpy()
        n_wo = int((crop["counts"] > full["gripper_num"]).sum())
        print("batch %d: %.1f s, valid crops %d / %d (%d drawn without replacement), class-1 %d, score-kept %d, keep3 %s" % (
            bi, time.time() - t0, len(crop["valid"]), next_grasp.shape[0], n_wo, len(sel_class), len(sel_score), rec["keep3"]))
        assert len(crop["valid"]) >= next_grasp.shape[0] // 2 and len(sel_class) > 0

    np.savez_compressed(os.path.join(HERE, "s9_refine_b8.npz"), **out_npz)
    with open(os.path.join(HERE, "s9_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for fn in ("s9_refine_b8.npz", "s9_meta.json"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


if __name__ == "__main__":
    main()
