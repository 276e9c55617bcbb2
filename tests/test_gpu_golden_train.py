"""The reference's TRAINING fixture held directly against the HIP path (one hop, not two).

``tests/golden/s4_train.npz`` / ``s4_meta.json`` were captured from the reference's own Python graph
(train.py:347-384: ScoreNet with labels -> get_grasp_allobj with grasp pickles -> GripperRegionNetwork with labels ->
stage-2 + refine losses; generator tests/golden/make_golden_train.py, run in the build container over the CPU oracle).
``tests/test_golden_train_cpu.py`` holds the host-side mirror to it on the CPU; the GPU training tests of
``tests/test_gpu_train.py`` compare with that mirror.  Here the same fixture is compared with what the MI355X
kernels produce: losses, the decoded stage-2 grasps, the label rows, the kept centres and the position of numpy's
global stream afterwards.

The one thing a GPU run cannot share with the fixture is torch's CPU random generator: the segmentation head applies
dropout(0.5) in training mode (pointnet2.py:78, nn/modules/mlp.py:99-101) and the fixture's masks were drawn from
``torch.manual_seed(cfg['torch_seed'])`` on the CPU.  The test draws the SAME masks the same way (CPU generator, same
call order and shapes: the mask of a CPU dropout is ``empty_like(x).bernoulli_(1 - p) / (1 - p)``, independent of the
values) and multiplies them in on the device.
"""
import contextlib
import io

import numpy as np
import pytest
import torch

from . import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, want, tol):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if w is None:
            assert g is None
        elif np.isnan(w):
            assert np.isnan(float(g))
        else:
            assert abs(float(g) - w) <= tol * max(1.0, abs(w)), (float(g), w)


def test_s4c_reference_training_forward_on_hip(monkeypatch):
    from regnet_for_3d_grasping_amd import synthetic
    from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
    from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
    from regnet_for_3d_grasping_amd.pn2_utils.nn import blocks
    from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
    m = gu.meta_train()
    exp = gu.load("s4_train.npz")
    cfg = m["cfg"]
    B, N = cfg["B"], cfg["N"]
    pc = synthetic.make_batch(cfg["scene_seed"], B, N)
    records = [synthetic.make_grasp_labels(pc[b].numpy(), cfg["label_seed"] + b) for b in range(B)]
    pc_score = torch.from_numpy(np.random.default_rng(cfg["label_seed"]).uniform(0, 1, (B, N)).astype(np.float32))
    net = ScoreNetwork(training=True)
    net.load_state_dict(synthetic.seeded_state_dict(net, cfg["score_weights_seed"]))
    rnet = GripperRegionNetwork(training=True, group_num=cfg["params"][2], gripper_num=cfg["gripper_num"],
                                grasp_score_threshold=cfg["grasp_score_threshold"], radius=cfg["gripper_params"][2],
                                reg_channel=cfg["reg_channel"])
    rnet.load_state_dict(synthetic.seeded_state_dict(rnet, cfg["region_weights_seed"]))
    net, rnet = net.to(DEV).train(), rnet.to(DEV).train()

    drawn = []
    torch_dropout = torch.nn.functional.dropout

    def cpu_mask_dropout(x, p=0.5, training=True, inplace=False):
        assert training and x.is_cuda and x.is_contiguous()
        mask = torch_dropout(torch.ones(x.shape, dtype=x.dtype), p=p, training=True)   # CPU generator
        drawn.append(tuple(x.shape))
        return x * mask.to(x.device)
    monkeypatch.setattr(blocks.F, "dropout", cpu_mask_dropout)

    torch.manual_seed(cfg["torch_seed"])
    np.random.seed(cfg["np_seed"])
    pc_d = pc.to(DEV)
    with contextlib.redirect_stdout(io.StringIO()):
        all_feature, score, loss = net(pc_d, pc_score.to(DEV), None)
        grouped = get_grasp_allobj(pc_d, score, cfg["params"], records)
        res = rnet(grouped[3], grouped[5], grouped[2], grouped[4], grouped[0], grouped[1], pc_d, all_feature,
                   cfg["gripper_params"], grouped[6], records)
    monkeypatch.undo()
    assert len(drawn) == 4 and all(s == (B, c, N) for s, c in zip(drawn, (512, 256, 256, 128))), drawn
    want = m["s4c"]
    # ScoreNet in training mode (batch statistics, the fixture's dropout masks): fp32 on both sides, different summation orders
    assert abs(float(loss) - want["score_loss"]) < 1e-5
    score_err = float(np.abs(score.detach()[:, ::16].cpu().numpy() - exp["s4c_score_sample"]).max())
    assert score_err <= 1e-4, score_err
    # everything from here on is a function of WHICH points score above 0.5: the same set as the reference's
    assert [int(v) for v in (score > 0.5).sum(1)] == want["positives"]
    np.testing.assert_allclose(grouped[6].cpu().numpy(), exp["s4c_labels"], rtol=0, atol=1e-5)   # label rows
    assert gu.sha(res[2].long()) == want["true_mask_sha256"]                                          # kept centres
    assert [int(k) for k in res[1]] == want["keep2"]
    _close(res[3], want["stage2_loss_tuple"], tol=1e-4)
    _close(res[4], want["stage2_correct"], tol=0)
    # the decoded grasps of the labelled centres: outputs of a head whose five BatchNorms normalise by the statistics of a
    # 128-ROW batch (B * 64 centres, training mode) -- a channel whose 128 values are close together divides a 1e-5 input
    # difference by a small standard deviation.  Measured 5.0e-4 on 29 of 530 values (the rest within 1e-4); the losses
    # above, which average over the rows, hold 1e-4.  tests/test_gpu_heads_train.py holds the head kernels themselves to
    # float64 (never more than 3 x torch's own fp32 error on the same rows).
    grasp_err = np.abs(res[0].cpu().numpy() - exp["s4c_next_grasp"])
    assert float(grasp_err.max()) <= 1e-3 and float(np.median(grasp_err)) <= 2e-5, (float(grasp_err.max()), float(np.median(grasp_err)))
    assert (len(res[13]) > 2) == want["refine_ran"]
    if want["refine_ran"]:
        _close(res[13], want["refine_loss_tuple"], tol=1e-4)
    total = loss.sum() + res[3][0].sum() + (res[13][0].sum() if len(res[13]) > 2 else 0.0)
    assert abs(float(total) - want["total_loss"]) <= 1e-4 * max(1.0, abs(want["total_loss"]))
    assert int(np.random.randint(0, 2 ** 31 - 1)) == want["np_state_after"]       # numpy's stream: consumed identically
    total.backward()
    assert net.extrat_featurePN2.sa_modules[0].mlp[0].conv.weight.grad is not None
    assert rnet.extrat_feature_region.conv.weight.grad is not None
    assert rnet.extrat_feature_region.linear_cls.weight.grad is None               # never used, like the reference
    print("S4c on HIP: score loss %.8f (ref %.8f), total %.6f (ref %.6f), score |err| %.2e, grasp |err| max %.2e median %.2e" % (
        float(loss), want["score_loss"], float(total), want["total_loss"], score_err, float(grasp_err.max()),
        float(np.median(grasp_err))))


def test_s4a_reference_loss_functions_on_hip():
    """S4a: the two loss functions on their own seeded inputs, fused kernels (csrc/losses.hip) vs the reference's numbers."""
    from regnet_for_3d_grasping_amd import region_losses, synthetic
    from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
    m = gu.meta_train()
    exp = gu.load("s4_train.npz")
    cfg = m["cfg"]
    net = GripperRegionNetwork(training=True, group_num=cfg["params"][2], gripper_num=cfg["gripper_num"],
                               grasp_score_threshold=cfg["grasp_score_threshold"], radius=cfg["gripper_params"][2],
                               reg_channel=cfg["reg_channel"])
    net.load_state_dict(synthetic.seeded_state_dict(net, cfg["region_weights_seed"]))
    net = net.to(DEV)
    stage2, refine = gu.loss_inputs(cfg["loss_inputs_seed"])
    tmpl = net.templates.float().reshape(-1, 4).to(DEV).contiguous()
    np.random.seed(cfg["np_seed"])
    assert region_losses.usable_stage2(stage2["first_grasp"].to(DEV), stage2["first_cls"].to(DEV), stage2["centres"].to(DEV),
                                       stage2["ground"].to(DEV))
    ng, lt, ct, next_gt, _, gmask = region_losses.stage2_loss(stage2["first_grasp"].to(DEV), stage2["first_cls"].to(DEV),
                                                              stage2["centres"].to(DEV), tmpl, stage2["ground"].to(DEV),
                                                              net.radius)
    _close(lt, m["s4a_stage2"]["loss_tuple"], tol=1e-5)
    _close(ct, m["s4a_stage2"]["correct"], tol=0)
    assert gu.sha(gmask.long()) == m["s4a_stage2"]["gmask_sha256"]
    np.testing.assert_allclose(ng.cpu().numpy(), exp["s4a_next_grasp"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(next_gt.cpu().numpy(), exp["s4a_next_gt"], rtol=0, atol=0)
    np.random.seed(cfg["np_seed"] + 1)
    r = net.compute_loss_refine(refine["next_grasp"].to(DEV), refine["next_x_cls"].to(DEV), refine["next_x_reg"].to(DEV),
                                refine["next_gt"].to(DEV))
    _close(r[5], m["s4a_refine"]["loss_tuple"], tol=1e-5)
    _close(r[6], m["s4a_refine"]["correct"], tol=0)
    assert gu.sha(r[3].long()) == m["s4a_refine"]["class_select_sha256"]
    assert gu.sha(r[4].long()) == m["s4a_refine"]["score_select_sha256"]
    np.testing.assert_allclose(r[0].cpu().numpy(), exp["s4a_select_class"], rtol=0, atol=2e-6)
