"""The oracle-backed mirror against the FULL-SIZE reference fixtures (tests/golden/make_golden_fullsize.py):
25 600- and 51 200-point ScoreNet forwards and the 25 600-point region stage, exact indices, floats to 1e-5.
This pins the oracle + mirror at the sizes BASELINE.json quotes; the GPU twin is tests/test_gpu_golden_fullsize.py."""
import contextlib
import io

import numpy as np
import torch

from . import golden_util as gu


def test_s7a_scorenet_25600_scene0_matches_reference(oracle_backend, monkeypatch):
    from regnet_for_3d_grasping_amd import synthetic
    m = gu.meta_full()
    exp = gu.load("s7a_scorenet_25600.npz")
    a = m["cfg"]["a"]
    rec = gu.OpRecorder(monkeypatch, oracle_backend)
    net = gu.build_scorenet_full(m)
    pc = synthetic.make_batch(a["scene_seed"], a["B"], a["N"])[:1].contiguous()   # configs[0]: one scene, batch 1
    with torch.no_grad():
        all_feature, score, _ = net(pc)
    gu.check_ops_per_scene(rec.log, m["s7a_ops"], [0])
    np.testing.assert_allclose(score.numpy(), exp["score"][:1], rtol=0, atol=1e-6)
    np.testing.assert_allclose(all_feature[:, ::m["cfg"]["feature_stride_a"], :].numpy(), exp["feature_sample"][:1],
                               rtol=0, atol=1e-5)
    assert int((score > 0.5).sum()) == m["s7a_positive"][0]


def test_s7b_scorenet_51200_matches_reference(oracle_backend, monkeypatch):
    from regnet_for_3d_grasping_amd import synthetic
    m = gu.meta_full()
    exp = gu.load("s7b_scorenet_51200.npz")
    b = m["cfg"]["b"]
    rec = gu.OpRecorder(monkeypatch, oracle_backend)
    net = gu.build_scorenet_full(m)
    with torch.no_grad():
        all_feature, score, _ = net(synthetic.make_batch(b["scene_seed"], b["B"], b["N"]))
    gu.check_ops_per_scene(rec.log, m["s7b_ops"], [0])
    np.testing.assert_allclose(score.numpy(), exp["score"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(all_feature[:, ::m["cfg"]["feature_stride_b"], :].numpy(), exp["feature_sample"],
                               rtol=0, atol=1e-5)


def test_s7c_region_stage_25600_matches_reference(oracle_backend):
    from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
    m = gu.meta_full()
    cfg = m["cfg"]
    exp = gu.load("s7c_region_25600.npz")
    pc, score, feat = gu.region_inputs_full(m)
    np.random.seed(cfg["c"]["np_seed"])
    center_pc, center_idx, g_idx, g, gm_idx, gm, _ = get_grasp_allobj(pc, score, cfg["params"], [])
    np.testing.assert_array_equal(center_idx.numpy(), exp["center_pc_index"])
    assert gu.sha(center_pc.float()) == m["s7c_group"]["center_pc_sha256"]
    assert gu.sha(g_idx.long()) == m["s7c_group"]["pc_group_index_sha256"]
    assert gu.sha(gm_idx.long()) == m["s7c_group"]["pc_group_more_index_sha256"]
    assert int(np.random.randint(0, 2 ** 31 - 1)) == m["s7c_group"]["np_state_after"]
    net = gu.build_regionnet_full(m)
    np.random.seed(cfg["c"]["np_seed"] + 1)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        out = net(g, gm, g_idx, gm_idx, center_pc, center_idx, pc, feat, cfg["gripper_params"], None, [])
    np.testing.assert_allclose(out[0].numpy(), exp["next_grasp"], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(out[2].numpy(), exp["true_mask"])
    assert [int(k) for k in out[1]] == m["s7c"]["keep2"]
    assert [int(k) for k in out[9]] == m["s7c"]["keep3"]
    assert (out[6] is not None) == m["s7c"]["refine_ran"]
    assert int(np.random.randint(0, 2 ** 31 - 1)) == m["s7c"]["np_state_after"]
