"""HIP path against the FULL-SIZE reference fixtures (tests/golden/make_golden_fullsize.py: the reference's own
Python graph over the oracle at 25 600 / 51 200 points): exact indices, floats within 1e-4 (north_star's tolerance).
Covers BASELINE.json configs[0]-[2] at their real sizes, configs[1] through the production pipeline."""
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


def _scenes_a(m):
    from regnet_for_3d_grasping_amd import synthetic
    a = m["cfg"]["a"]
    return synthetic.make_batch(a["scene_seed"], a["B"], a["N"]).to(DEV)


def test_s7a_scorenet_batch4_25600_fused(monkeypatch):
    import regnet_for_3d_grasping_amd.pn2_utils.function as fn
    m = gu.meta_full()
    exp = gu.load("s7a_scorenet_25600.npz")
    rec = gu.OpRecorder(monkeypatch, fn.pn2_ext)
    net = gu.build_scorenet_full(m, DEV)
    with torch.no_grad():
        all_feature, score, _ = net(_scenes_a(m))
    gu.check_ops_per_scene(rec.log, m["s7a_ops"], range(m["cfg"]["a"]["B"]))
    err = float(np.abs(score.cpu().numpy() - exp["score"]).max())
    print("s7a score max abs err vs reference: %.3e" % err)
    np.testing.assert_allclose(score.cpu().numpy(), exp["score"], **TOL)
    np.testing.assert_allclose(all_feature[:, ::m["cfg"]["feature_stride_a"], :].cpu().numpy(), exp["feature_sample"], **TOL_FEATURE)


def test_s7a_single_scene_batch1(monkeypatch):
    """configs[0]'s shape on the GPU: one 25 600-point scene, batch 1."""
    import regnet_for_3d_grasping_amd.pn2_utils.function as fn
    m = gu.meta_full()
    exp = gu.load("s7a_scorenet_25600.npz")
    rec = gu.OpRecorder(monkeypatch, fn.pn2_ext)
    net = gu.build_scorenet_full(m, DEV)
    with torch.no_grad():
        all_feature, score, _ = net(_scenes_a(m)[2:3].contiguous())
    gu.check_ops_per_scene(rec.log, m["s7a_ops"], [2])
    np.testing.assert_allclose(score.cpu().numpy(), exp["score"][2:3], **TOL)


def test_s7a_config1_score_only_pipeline():
    """BASELINE.json configs[1] -- ScoreNet forward, batch 4 x 25 600, no region stage -- through
    ``ForwardPipeline(with_region=False)`` (what ``bench.py --score-only --batch 4`` times): three batches in flight,
    every one of them must reproduce the reference's scores and features."""
    from regnet_for_3d_grasping_amd import pipeline
    m = gu.meta_full()
    exp = gu.load("s7a_scorenet_25600.npz")
    net = gu.build_scorenet_full(m, DEV)
    _, region_net = pipeline.build_models(DEV)
    pipe = pipeline.ForwardPipeline(net, region_net, with_region=False)
    pc = _scenes_a(m)
    order = [[0, 1, 2, 3], [3, 2, 1, 0], [1, 0, 3, 2]]
    batches = [pc[o].contiguous() for o in order]
    outs = list(pipe.run(iter(batches)))
    torch.cuda.synchronize()
    assert len(outs) == 3
    stride = m["cfg"]["feature_stride_a"]
    for o, out in zip(order, outs):
        np.testing.assert_allclose(out["score"].cpu().numpy(), exp["score"][o], **TOL)
        np.testing.assert_allclose(out["all_feature"][:, ::stride, :].cpu().numpy(), exp["feature_sample"][o], **TOL_FEATURE)


def test_s7b_scorenet_51200(monkeypatch):
    import regnet_for_3d_grasping_amd.pn2_utils.function as fn
    from regnet_for_3d_grasping_amd import synthetic
    m = gu.meta_full()
    exp = gu.load("s7b_scorenet_51200.npz")
    b = m["cfg"]["b"]
    rec = gu.OpRecorder(monkeypatch, fn.pn2_ext)
    net = gu.build_scorenet_full(m, DEV)
    with torch.no_grad():
        all_feature, score, _ = net(synthetic.make_batch(b["scene_seed"], b["B"], b["N"]).to(DEV))
    gu.check_ops_per_scene(rec.log, m["s7b_ops"], [0])
    np.testing.assert_allclose(score.cpu().numpy(), exp["score"], **TOL)
    np.testing.assert_allclose(all_feature[:, ::m["cfg"]["feature_stride_b"], :].cpu().numpy(), exp["feature_sample"], **TOL_FEATURE)


def test_s7c_region_stage_25600():
    from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
    m = gu.meta_full()
    cfg = m["cfg"]
    exp = gu.load("s7c_region_25600.npz")
    pc, score, feat = gu.region_inputs_full(m, DEV)
    np.random.seed(cfg["c"]["np_seed"])
    center_pc, center_idx, g_idx, g, gm_idx, gm, _ = get_grasp_allobj(pc, score, cfg["params"], [])
    np.testing.assert_array_equal(center_idx.cpu().numpy(), exp["center_pc_index"])
    assert gu.sha(center_pc.float()) == m["s7c_group"]["center_pc_sha256"]
    assert gu.sha(g_idx.long()) == m["s7c_group"]["pc_group_index_sha256"]
    assert gu.sha(gm_idx.long()) == m["s7c_group"]["pc_group_more_index_sha256"]
    assert int(np.random.randint(0, 2 ** 31 - 1)) == m["s7c_group"]["np_state_after"]
    net = gu.build_regionnet_full(m, DEV)
    np.random.seed(cfg["c"]["np_seed"] + 1)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        out = net(g, gm, g_idx, gm_idx, center_pc, center_idx, pc, feat, cfg["gripper_params"], None, [])
    np.testing.assert_allclose(out[0].cpu().numpy(), exp["next_grasp"], **TOL)
    np.testing.assert_array_equal(out[2].cpu().numpy(), exp["true_mask"])
    assert [int(k) for k in out[1]] == m["s7c"]["keep2"]
