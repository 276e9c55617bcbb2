"""The oracle-backed mirror against S9 (tests/golden/make_golden_refine.py): BASELINE.json configs[2] -- ScoreNet, region
grouping, grasp-region head AND the refine stage, 8 x 25 600 points -- with a calibrated region head, so that the refine
network actually runs (~450 valid crops of 512, ~225 class-1 grasps per batch) instead of the "fewer than two valid crops"
no-op of S7 / S8.  Everything the REFERENCE's graph produced is reproduced on the CPU: exact indices (centres, groups,
valid crops, candidate counts, the 64 scene indices of every crop, selection masks), the numpy stream position, floats to
1e-5.  The GPU twin is tests/test_gpu_pipeline_b8.py::test_config2_refine_stage_runs_against_reference_fixtures."""
import contextlib
import io

import numpy as np
import torch

from . import golden_util as gu


def test_s9_refine_stage_batch8_matches_reference(oracle_backend, monkeypatch):
    from regnet_for_3d_grasping_amd import synthetic
    import regnet_for_3d_grasping_amd.gripper_region_network as grn
    from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
    m7, m9 = gu.meta_full(), gu.meta_refine()
    full, cfg = m7["cfg"], m9["cfg"]
    exp = gu.load("s9_refine_b8.npz")
    pc = synthetic.make_batch(cfg["scene_seed"], cfg["B"], cfg["N"])
    snet = gu.build_scorenet_full(m7)
    with torch.no_grad():
        feat, score, _ = snet(pc)
    np.testing.assert_allclose(score.numpy(), gu.load("s8_b8_25600.npz")["score"], rtol=0, atol=1e-6)
    rnet = gu.build_regionnet_refine(m7, m9)
    spy = gu.CropSpy(monkeypatch, grn)
    counts = []
    orig_choice_rows = grn.np_random.choice_rows

    def choice_rows(count, size, mode):
        if mode == 1:                            # the crop's draws (mode 0: the grouping's resampling)
            counts.append(np.asarray(count).copy())
        return orig_choice_rows(count, size, mode)
    monkeypatch.setattr(grn.np_random, "choice_rows", choice_rows)

    for bi, order in enumerate(cfg["orders"]):
        b, p = m9["batches"][bi], "b%d_" % bi
        idx = torch.tensor(order)
        pcb, scb, ftb = pc[idx].contiguous(), score[idx].contiguous(), feat[idx].contiguous()
        np.random.seed(cfg["np_seed"] + bi)
        center_pc, center_idx, g_idx, g, gm_idx, gm, _ = get_grasp_allobj(pcb, scb, full["params"], [])
        np.testing.assert_array_equal(center_idx.numpy(), exp[p + "center_pc_index"])
        assert gu.sha(center_pc.float()) == b["center_pc_sha256"]
        assert gu.sha(g_idx.long()) == b["pc_group_index_sha256"]
        assert gu.sha(gm_idx.long()) == b["pc_group_more_index_sha256"]
        assert int(np.random.get_state()[2]) == b["np_state_after_grouping"]
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            out = rnet(g, gm, g_idx, gm_idx, center_pc, center_idx, pcb, ftb, full["gripper_params"], None, [])
        assert int(np.random.randint(0, 2 ** 31 - 1)) == b["np_draw_after"], "numpy stream position after batch %d" % bi
        np.testing.assert_allclose(out[0].numpy(), exp[p + "next_grasp"], rtol=0, atol=1e-5)
        np.testing.assert_array_equal(out[2].numpy(), exp[p + "true_mask"])
        call = spy.calls[bi]
        np.testing.assert_array_equal(call["valid"].numpy(), exp[p + "crop_valid"])
        np.testing.assert_array_equal(counts[bi][exp[p + "crop_valid"]], exp[p + "crop_counts"])
        np.testing.assert_array_equal(call["index_inall"].numpy(), exp[p + "crop_index_inall"])
        assert gu.sha(call["index_inall"].long()) == b["crop_index_inall_sha256"]
        assert b["refine_ran"] and out[6] is not None
        np.testing.assert_array_equal(out[11].numpy(), exp[p + "final_mask"])
        np.testing.assert_array_equal(out[12].numpy(), exp[p + "final_mask_sthre"])
        np.testing.assert_allclose(out[6].numpy(), exp[p + "select_grasp_class"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(out[7].numpy(), exp[p + "select_grasp_score"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(out[8].numpy(), exp[p + "select_grasp_class_stage2"], rtol=0, atol=1e-5)
        assert [int(k) for k in out[1]] == b["keep2"]
        assert [int(k) for k in out[9]] == b["keep3"] and sum(b["keep3"]) > 0
        assert [int(k) for k in out[10]] == b["keep3_score"]
        assert b["valid_crops"] >= 256          # the calibration's purpose: at least half of the 512 crops are valid
