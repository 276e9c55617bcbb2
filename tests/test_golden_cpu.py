"""The host-side mirror (modules / models / grouping) driven by the CPU oracle must reproduce
the fixtures generated from the REFERENCE's Python graph (tests/golden/make_golden.py):
exact indices, floats to 1e-6.  Runs without a GPU."""
import contextlib
import io

import numpy as np
import torch

from . import golden_util as gu


def test_s1_scorenet_matches_reference(oracle_backend, monkeypatch):
    m = gu.meta()
    exp = gu.load("s1_scorenet.npz")
    rec = gu.OpRecorder(monkeypatch, oracle_backend)
    net = gu.build_scorenet(m)
    pc = gu.scenes(m["cfg"])
    with torch.no_grad():
        all_feature, score, loss = net(pc)
    assert loss is None
    rec.check_against(m["s1_ops"])
    assert tuple(all_feature.shape) == (m["cfg"]["B"], m["cfg"]["N"], 256)
    np.testing.assert_allclose(score.numpy(), exp["score"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(all_feature[:, ::64, :].numpy(), exp["feature_sample"], rtol=0, atol=1e-5)
    assert [int(v) for v in (score > 0.5).sum(1)] == m["s1_positive"]


def _s2(m, pc):
    from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
    cfg = m["cfg"]
    pscore = gu.pseudo_scores(cfg["s2_score_seed"], cfg["B"], cfg["N"]).to(pc.device)
    np.random.seed(cfg["s2_np_seed"])
    return get_grasp_allobj(pc, pscore, cfg["params"], [])


def test_s2_region_grouping_matches_reference(oracle_backend):
    m = gu.meta()
    exp = gu.load("s2_grouping.npz")
    pc = gu.scenes(m["cfg"])
    center_pc, center_idx, g_idx, g, gm_idx, gm, labels = _s2(m, pc)
    assert labels is None
    np.testing.assert_array_equal(center_idx.numpy(), exp["center_pc_index"])
    assert gu.sha(center_pc.float()) == m["s2"]["center_pc_sha256"]
    assert gu.sha(g_idx.long()) == m["s2"]["pc_group_index_sha256"]
    assert gu.sha(g.float()) == m["s2"]["pc_group_sha256"]
    assert gu.sha(gm_idx.long()) == m["s2"]["pc_group_more_index_sha256"]
    assert gu.sha(gm.float()) == m["s2"]["pc_group_more_sha256"]
    # the numpy RNG must have been consumed exactly as by the reference
    assert int(np.random.randint(0, 2 ** 31 - 1)) == m["s2"]["np_state_after"]


def test_s3_region_network_matches_reference(oracle_backend):
    from regnet_for_3d_grasping_amd.gripper_region_network import get_gripper_region_transform
    m = gu.meta()
    cfg = m["cfg"]
    exp = gu.load("s3_region.npz")
    pc = gu.scenes(cfg)
    center_pc, center_idx, g_idx, g, gm_idx, gm, _ = _s2(m, pc)
    feat = gu.pseudo_feature(cfg["s3_feature_seed"], cfg["B"], cfg["N"])
    net = gu.build_regionnet(m)
    np.random.seed(cfg["s3_np_seed"])
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        out = net(g, gm, g_idx, gm_idx, center_pc, center_idx, pc, feat, cfg["gripper_params"], None, [])
    (next_grasp, keep2, true_mask, _, _, _, sel_class, sel_score, sel_class_s2, keep3, keep3s, final_mask,
     final_mask_sthre, _, _, _) = out
    np.testing.assert_allclose(next_grasp.numpy(), exp["next_grasp"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(true_mask.numpy(), exp["true_mask"])
    assert [int(k) for k in keep2] == m["s3"]["keep2"]
    assert m["s3"]["refine_ran"] == (sel_class is not None)
    np.testing.assert_array_equal(final_mask.numpy(), exp["final_mask"])
    np.testing.assert_array_equal(final_mask_sthre.numpy(), exp["final_mask_sthre"])
    np.testing.assert_allclose(sel_class.numpy(), exp["select_grasp_class"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(sel_score.numpy(), exp["select_grasp_score"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(sel_class_s2.numpy(), exp["select_grasp_class_stage2"], rtol=0, atol=1e-6)
    assert [int(k) for k in keep3] == m["s3"]["keep3"]
    assert [int(k) for k in keep3s] == m["s3"]["keep3_score"]

    # crop stage on its own, teacher-forced from the golden stage-2 grasps
    B = cfg["B"]
    np.random.seed(cfg["s3_np_seed"])
    gp, gidx, gidx_all, gmask = get_gripper_region_transform(
        gm[:, :, :, :6].clone().view(B * 64, -1, 6), gm_idx.view(B * 64, -1), torch.from_numpy(exp["next_grasp"]),
        cfg["gripper_num"], cfg["gripper_params"])
    np.testing.assert_array_equal(gmask.numpy(), exp["crop_valid"])
    np.testing.assert_array_equal(gidx_all.numpy(), exp["crop_index_inall"])
    assert gu.sha(gidx.long()) == m["s3"]["crop_index_sha256"]
    assert gu.sha(gp) == m["s3"]["crop_pc_sha256"]
