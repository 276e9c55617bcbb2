"""GPU path against the committed golden fixtures (generated from the reference's Python graph):
exact indices, floats within 1e-4 (the tolerance BASELINE.json's north_star states)."""
import contextlib
import io

import numpy as np
import pytest
import torch

from . import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# north_star's bound is ABSOLUTE: outputs (scores, grasp tuples, class-1 grasps) within 1e-4 of the reference, no relative
# slack.  The 256-channel feature map is an intermediate with magnitudes up to ~10: it keeps a bound relative to its size.
TOL = dict(rtol=0.0, atol=1e-4)
TOL_FEATURE = dict(rtol=1e-4, atol=1e-4)


def test_s1_scorenet_on_gpu(monkeypatch):
    import regnet_for_3d_grasping_amd.pn2_utils.function as fn
    m = gu.meta()
    exp = gu.load("s1_scorenet.npz")
    rec = gu.OpRecorder(monkeypatch, fn.pn2_ext)
    net = gu.build_scorenet(m, DEV)
    pc = gu.scenes(m["cfg"], DEV)
    with torch.no_grad():
        all_feature, score, _ = net(pc)
    if rec.log:  # op-granular path: every index tensor must match the reference bit for bit
        rec.check_against(m["s1_ops"])
    np.testing.assert_allclose(score.cpu().numpy(), exp["score"], **TOL)
    np.testing.assert_allclose(all_feature[:, ::64, :].cpu().numpy(), exp["feature_sample"], **TOL_FEATURE)


def test_s1_scorenet_unfused_ops_on_gpu(monkeypatch):
    """Same network forced through the operator-granular path (fused dispatch off)."""
    import regnet_for_3d_grasping_amd.fused as fused
    import regnet_for_3d_grasping_amd.pn2_utils.function as fn
    monkeypatch.setattr(fused, "ENABLED", False)
    m = gu.meta()
    exp = gu.load("s1_scorenet.npz")
    rec = gu.OpRecorder(monkeypatch, fn.pn2_ext)
    net = gu.build_scorenet(m, DEV)
    with torch.no_grad():
        all_feature, score, _ = net(gu.scenes(m["cfg"], DEV))
    rec.check_against(m["s1_ops"])
    np.testing.assert_allclose(score.cpu().numpy(), exp["score"], **TOL)
    np.testing.assert_allclose(all_feature[:, ::64, :].cpu().numpy(), exp["feature_sample"], **TOL_FEATURE)


def _s2(m, pc):
    from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
    cfg = m["cfg"]
    pscore = gu.pseudo_scores(cfg["s2_score_seed"], cfg["B"], cfg["N"]).to(pc.device)
    np.random.seed(cfg["s2_np_seed"])
    return get_grasp_allobj(pc, pscore, cfg["params"], [])


def test_s2_region_grouping_on_gpu():
    m = gu.meta()
    exp = gu.load("s2_grouping.npz")
    pc = gu.scenes(m["cfg"], DEV)
    center_pc, center_idx, g_idx, g, gm_idx, gm, _ = _s2(m, pc)
    np.testing.assert_array_equal(center_idx.cpu().numpy(), exp["center_pc_index"])
    assert gu.sha(center_pc.float()) == m["s2"]["center_pc_sha256"]
    assert gu.sha(g_idx.long()) == m["s2"]["pc_group_index_sha256"]
    assert gu.sha(g.float()) == m["s2"]["pc_group_sha256"]
    assert gu.sha(gm_idx.long()) == m["s2"]["pc_group_more_index_sha256"]
    assert gu.sha(gm.float()) == m["s2"]["pc_group_more_sha256"]
    assert int(np.random.randint(0, 2 ** 31 - 1)) == m["s2"]["np_state_after"]


def test_s3_region_network_on_gpu():
    from regnet_for_3d_grasping_amd.gripper_region_network import get_gripper_region_transform
    m = gu.meta()
    cfg = m["cfg"]
    exp = gu.load("s3_region.npz")
    pc = gu.scenes(cfg, DEV)
    center_pc, center_idx, g_idx, g, gm_idx, gm, _ = _s2(m, pc)
    feat = gu.pseudo_feature(cfg["s3_feature_seed"], cfg["B"], cfg["N"]).to(DEV)
    net = gu.build_regionnet(m, DEV)
    np.random.seed(cfg["s3_np_seed"])
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        out = net(g, gm, g_idx, gm_idx, center_pc, center_idx, pc, feat, cfg["gripper_params"], None, [])
    next_grasp, keep2, true_mask = out[0], out[1], out[2]
    np.testing.assert_allclose(next_grasp.cpu().numpy(), exp["next_grasp"], **TOL)
    np.testing.assert_array_equal(true_mask.cpu().numpy(), exp["true_mask"])
    assert [int(k) for k in keep2] == m["s3"]["keep2"]

    # crop + refine, teacher-forced from the golden stage-2 grasps so fp32 noise in next_grasp
    # cannot move a point across a box face
    B = cfg["B"]
    np.random.seed(cfg["s3_np_seed"])
    gp, gidx, gidx_all, gmask = get_gripper_region_transform(
        gm[:, :, :, :6].clone().view(B * 64, -1, 6), gm_idx.view(B * 64, -1),
        torch.from_numpy(exp["next_grasp"]).to(DEV), cfg["gripper_num"], cfg["gripper_params"])
    np.testing.assert_array_equal(gmask.cpu().numpy(), exp["crop_valid"])
    np.testing.assert_array_equal(gidx_all.cpu().numpy(), exp["crop_index_inall"])
    assert gu.sha(gidx.long()) == m["s3"]["crop_index_sha256"]

    # full forward: with identical crops the refine outputs must agree within tolerance
    sel_class, sel_score, sel_class_s2, final_mask, final_mask_sthre = out[6], out[7], out[8], out[11], out[12]
    if final_mask is not None and np.array_equal(final_mask.cpu().numpy(), exp["final_mask"]):
        np.testing.assert_allclose(sel_class.cpu().numpy(), exp["select_grasp_class"], **TOL)
        np.testing.assert_allclose(sel_class_s2.cpu().numpy(), exp["select_grasp_class_stage2"], **TOL)
        np.testing.assert_array_equal(final_mask_sthre.cpu().numpy(), exp["final_mask_sthre"])
    else:
        pytest.fail("refine selection differs from the golden run")


def test_inference_script_scale_grouping():
    """test.py-scale region stage (test.py:68-71: 4000 centres, 256 / 2048-point groups) against the
    oracle-backed mirror on the same scores: exact indices, same numpy stream."""
    from oracle.install import oracle_backend
    from regnet_for_3d_grasping_amd import synthetic
    from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
    N = 12000
    pc = synthetic.make_batch(9000, 1, N)
    rng = np.random.default_rng(4)
    score = torch.from_numpy(rng.uniform(0, 1, (1, N)).astype(np.float32))
    params = [4000, 0.5, 256, 0.1, 2048, 0.8, 0.08, 0.01, 0.06]
    with oracle_backend():
        np.random.seed(21)
        want = get_grasp_allobj(pc, score, params, [])
        after = int(np.random.randint(0, 2 ** 31 - 1))
    np.random.seed(21)
    got = get_grasp_allobj(pc.to(DEV), score.to(DEV), params, [])
    assert int(np.random.randint(0, 2 ** 31 - 1)) == after
    for g, w in zip(got[:6], want[:6]):
        assert tuple(g.shape) == tuple(w.shape)
        assert torch.equal(g.cpu(), w)


def test_scorenet_dense_scene_51200_points():
    """BASELINE.json configs[4] scene size (51 200 points: level-1 FPS takes the streaming kernel,
    every other kernel a larger grid): fused forward vs the oracle-backed mirror on the CPU."""
    from oracle.install import oracle_backend
    from regnet_for_3d_grasping_amd import pipeline, synthetic
    pc = synthetic.make_batch(9100, 1, 51200)
    cpu_net, _ = pipeline.build_models("cpu")
    with oracle_backend():
        synthetic.calibrate_score_head(cpu_net, pc)
        with torch.no_grad():
            feat_ref, score_ref, _ = cpu_net(pc)
    gpu_net, _ = pipeline.build_models(DEV)
    gpu_net.load_state_dict(cpu_net.state_dict())
    with torch.no_grad():
        feat, score, _ = gpu_net(pc.to(DEV))
    assert tuple(feat.shape) == (1, 51200, 256)
    np.testing.assert_allclose(score.cpu().numpy(), score_ref.numpy(), **TOL)
    np.testing.assert_allclose(feat[:, ::97, :].cpu().numpy(), feat_ref[:, ::97, :].numpy(), **TOL_FEATURE)
