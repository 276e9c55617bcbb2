"""BASELINE.json configs[2] AT ITS OWN BATCH SIZE through the bench path: ``ForwardPipeline`` (default grouping: the
level-1..3 sampling of all three batches shares one launch per level) over three batches of 8 x 25 600 points, against
fixtures the REFERENCE's own Python graph produced end to end at B=8 (tests/golden/make_golden_b8.py -> s8_*).

Checked per batch: every FPS / ball-query / 3-NN index tensor of every scene (SHA-256, bit-exact), the scores (absolute
1e-4, north_star's bound), a strided sample of the 256-channel feature map; and the region stage INSIDE the pipeline
(its worker thread, its streams), teacher-forced with the reference's scores so that centre selection sees the same
positives: centres and both group index tensors bit-exact, numpy's stream position after the heads' crop draws, grasp
tuples within 1e-4 absolute, valid-crop mask and per-scene keep counts equal."""
import json
import os
import threading

import numpy as np
import pytest
import torch

from . import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL = 1e-4          # absolute, no relative slack (BASELINE.json north_star: "outputs within 1e-4 of reference")
FEATURE_TOL = dict(rtol=1e-4, atol=1e-4)


def _meta8():
    with open(os.path.join(gu.GOLDEN, "s8_meta.json")) as f:
        return json.load(f)


class _Recorder:
    """Per-scene SHA-256 of the ScoreNet geometry ops, whatever batch grouping the pipeline launches them with."""
    LEVEL_M = (5120, 1024, 256)

    def __init__(self, monkeypatch, ext, scene_order):
        self.flat = scene_order                      # scene id of every row the pipeline will process, in order
        self.cursor = {}
        self.seen = {}                               # (op, shape[1:]) -> list of (scene, index sha, aux sha or None)
        self.lock = threading.Lock()
        for name in ("farthest_point_sample", "ball_query", "point_search"):
            monkeypatch.setattr(ext, name, self._wrap(name, getattr(ext, name)))

    def _wrap(self, name, orig):
        def wrapped(*a):
            out = orig(*a)
            if name == "farthest_point_sample" and a[1] not in self.LEVEL_M:
                return out                           # the region stage's centre picker (64 centres), worker thread
            outs = out if isinstance(out, (list, tuple)) else [out]
            key = (name, tuple(outs[0].shape[1:]))
            with self.lock:
                at = self.cursor.get(key, 0)
                self.cursor[key] = at + outs[0].shape[0]
                rows = self.seen.setdefault(key, [])
                for i in range(outs[0].shape[0]):
                    aux = gu.sha(outs[1][i]) if len(outs) > 1 else None   # ball-query counts / 3-NN squared distances
                    rows.append((self.flat[at + i], gu.sha(outs[0][i]), aux))
            return out
        return wrapped

    def check(self, expected_ops, rows_total):
        assert len(self.seen) == len(expected_ops) == 9
        for exp in expected_ops:
            rows = self.seen[(exp["op"], tuple(exp["shape"][1:]))]
            assert len(rows) == rows_total, (exp["op"], exp["shape"], len(rows))
            for scene, idx_sha, aux_sha in rows:
                assert idx_sha == exp["index_sha256"][scene], "%s %s: index mismatch, scene %d" % (exp["op"], exp["shape"], scene)
                if aux_sha is not None:
                    assert aux_sha == exp["aux_sha256"][scene], "%s %s: count mismatch, scene %d" % (exp["op"], exp["shape"], scene)


def _teacher_forced(ref_score, orders, np_seed):
    """``ForwardPipeline`` whose region stage of batch k sees the REFERENCE's scores of its scenes (the HIP scores, checked
    separately, are within 1e-4 of them, which moves a few of the 200 000 points across the 0.5 threshold) and a per-batch
    numpy seed (as the fixture generators)."""
    from regnet_for_3d_grasping_amd import np_random, pipeline

    class TeacherForced(pipeline.ForwardPipeline):
        n_region = 0
        draws = []

        def _region(self, item):
            k = TeacherForced.n_region
            TeacherForced.n_region += 1
            item["hip_score"] = item["score"]
            with torch.cuda.stream(self.s_reg):
                self.s_reg.wait_event(item["mlp_done"])
                item["score"] = ref_score[torch.tensor(orders[k], device=DEV)].contiguous()
            np.random.seed(np_seed + k)
            hip_score = item["hip_score"]
            out = super()._region(item)
            out["done"].synchronize()
            np_random.flush()               # the region stages of a run keep numpy's generator on the device
            TeacherForced.draws.append(int(np.random.randint(0, 2 ** 31 - 1)))
            out["hip_score"] = hip_score
            return out
    return TeacherForced


def test_config2_batch8_pipeline_against_reference_fixtures(monkeypatch):
    from regnet_for_3d_grasping_amd import synthetic
    import regnet_for_3d_grasping_amd.pn2_utils.function as fn
    m7, m8 = gu.meta_full(), _meta8()
    cfg = m8["cfg"]
    exp = gu.load("s8_b8_25600.npz")
    orders = cfg["orders"]
    ref_score = torch.from_numpy(exp["score"]).to(DEV)
    net = gu.build_scorenet_full(m7, DEV)
    rnet = gu.build_regionnet_full(m7, DEV)
    pc = synthetic.make_batch(cfg["scene_seed"], cfg["B"], cfg["N"]).to(DEV)
    rec = _Recorder(monkeypatch, fn.pn2_ext, [s for o in orders for s in o])

    TeacherForced = _teacher_forced(ref_score, orders, cfg["np_seed"])
    pipe = TeacherForced(net, rnet)                         # default grouping = the bench path
    outs = list(pipe.run(iter([pc[o].contiguous() for o in orders])))
    torch.cuda.synchronize()
    assert len(outs) == 3
    rec.check(m8["ops"], 3 * cfg["B"])
    stride = cfg["feature_stride"]
    worst = 0.0
    for k, (o, out) in enumerate(zip(orders, outs)):
        b = m8["batches"][k]
        err = float(np.abs(out["hip_score"].cpu().numpy() - exp["score"][o]).max())
        worst = max(worst, err)
        assert err <= ATOL, "batch %d: score max abs err %.3e" % (k, err)
        np.testing.assert_allclose(out["all_feature"][:, ::stride, :].cpu().numpy(), exp["feature_sample"][o], **FEATURE_TOL)
        np.testing.assert_array_equal(out["center_pc_index"].cpu().numpy(), exp["b%d_center_pc_index" % k])
        assert gu.sha(out["pc_group_index"].long()) == b["pc_group_index_sha256"]
        assert gu.sha(out["pc_group_more_index"].long()) == b["pc_group_more_index_sha256"]
        np.testing.assert_array_equal(out["true_mask"].cpu().numpy(), exp["b%d_true_mask" % k])
        assert [int(v) for v in out["keep_per_scene"]] == b["keep2"]
        np.testing.assert_allclose(out["next_grasp"].cpu().numpy(), exp["b%d_next_grasp" % k], rtol=0.0, atol=ATOL)
        want = exp["b%d_select_grasp_class" % k]
        got = out["select_grasp_class"]
        assert (0 if got is None else got.shape[0]) == want.shape[0]
        if want.shape[0]:
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0.0, atol=ATOL)
        assert TeacherForced.draws[k] == b["np_draw_after"], "numpy stream position after batch %d" % k
    print("configs[2] B=8 pipeline: score max abs err vs reference %.3e" % worst)


def test_scores_against_the_float64_evaluation():
    """How much of |HIP - reference| is whose rounding: the S8 scenes against a float64 evaluation of the same graph
    (tests/golden/make_fp64_truth.py -> s8_score_fp64.npz; same index tensors, every floating-point op in double).  The
    reference's own fp32 scores sit 4.1e-5 from it; the HIP path must stay within north_star's 1e-4 of it as well (measured
    6.7e-5; torch's own GPU convolutions on the operator-granular path: 9.4e-5 -- profiles/archive/r03_error_budget.txt)."""
    from regnet_for_3d_grasping_amd import synthetic
    m7, m8 = gu.meta_full(), _meta8()
    cfg = m8["cfg"]
    truth = np.load(os.path.join(gu.GOLDEN, "s8_score_fp64.npz"))
    net = gu.build_scorenet_full(m7, DEV)
    pc = synthetic.make_batch(cfg["scene_seed"], cfg["B"], cfg["N"]).to(DEV)
    with torch.no_grad():
        _, score, _ = net(pc)
    err = np.abs(score.cpu().numpy().astype(np.float64) - truth["score"])
    print("HIP vs float64 evaluation: max %.3e mean %.3e (reference vs float64: max %.3e)" % (
        err.max(), err.mean(), float(truth["reference_max_abs_err"].max())))
    assert err.max() <= ATOL


def test_config2_refine_stage_runs_against_reference_fixtures(monkeypatch):
    """configs[2] with its THIRD network running: S9 (tests/golden/make_golden_refine.py) -- the S8 scenes with a calibrated
    region head, ~450 valid crops of 512 and ~225 class-1 grasps per batch -- through ``ForwardPipeline`` (the bench
    path; the bench uses the same calibration).  Per batch: scores as S8; centres / groups bit-exact; the stage-2 grasps
    the HIP heads decoded within 1e-4 of the reference's; then, teacher-forced with the reference's stage-2 grasps (as
    test_s3: fp32 noise in a decoded frame must not move a point across a box face): valid crop ids, the 64 scene
    indices of every crop (array + SHA-256), numpy's stream position after the crop draws, ``final_mask`` /
    ``final_mask_sthre`` (the refine class / score selections), ``select_grasp_class / score`` within 1e-4, keep counts.
    (gripper_region_network.py:311-359, pointnet2.py:227-254.)"""
    from regnet_for_3d_grasping_amd import synthetic
    import regnet_for_3d_grasping_amd.gripper_region_network as grn
    m7, m9 = gu.meta_full(), gu.meta_refine()
    cfg = m9["cfg"]
    exp = gu.load("s9_refine_b8.npz")
    orders = cfg["orders"]
    ref_score = torch.from_numpy(gu.load("s8_b8_25600.npz")["score"]).to(DEV)
    net = gu.build_scorenet_full(m7, DEV)
    rnet = gu.build_regionnet_refine(m7, m9, DEV)
    pc = synthetic.make_batch(cfg["scene_seed"], cfg["B"], cfg["N"]).to(DEV)
    spy = gu.CropSpy(monkeypatch, grn, forced=[exp["b%d_next_grasp" % k] for k in range(len(orders))])
    TeacherForced = _teacher_forced(ref_score, orders, cfg["np_seed"])
    pipe = TeacherForced(net, rnet)
    outs = list(pipe.run(iter([pc[o].contiguous() for o in orders])))
    torch.cuda.synchronize()
    assert len(outs) == len(orders) == len(spy.calls)
    worst = 0.0
    for k, (o, out) in enumerate(zip(orders, outs)):
        b, p = m9["batches"][k], "b%d_" % k
        np.testing.assert_array_equal(out["center_pc_index"].cpu().numpy(), exp[p + "center_pc_index"])
        assert gu.sha(out["pc_group_index"].long()) == b["pc_group_index_sha256"]
        assert gu.sha(out["pc_group_more_index"].long()) == b["pc_group_more_index_sha256"]
        np.testing.assert_array_equal(out["true_mask"].cpu().numpy(), exp[p + "true_mask"])
        assert [int(v) for v in out["keep_per_scene"]] == b["keep2"]
        np.testing.assert_allclose(out["next_grasp"].cpu().numpy(), exp[p + "next_grasp"], rtol=0.0, atol=ATOL)
        np.testing.assert_allclose(spy.own[k].cpu().numpy(), exp[p + "next_grasp"], rtol=0.0, atol=ATOL)
        worst = max(worst, float(np.abs(spy.own[k].cpu().numpy() - exp[p + "next_grasp"]).max()))
        call = spy.calls[k]
        np.testing.assert_array_equal(call["valid"].cpu().numpy(), exp[p + "crop_valid"])
        np.testing.assert_array_equal(call["index_inall"].cpu().numpy(), exp[p + "crop_index_inall"])
        assert gu.sha(call["index_inall"].long()) == b["crop_index_inall_sha256"]
        assert TeacherForced.draws[k] == b["np_draw_after"], "numpy stream position after batch %d" % k
        assert out["select_grasp_class"] is not None and b["refine_ran"]
        np.testing.assert_array_equal(out["final_mask"].cpu().numpy(), exp[p + "final_mask"])
        np.testing.assert_allclose(out["select_grasp_class"].cpu().numpy(), exp[p + "select_grasp_class"], rtol=0.0, atol=ATOL)
        np.testing.assert_allclose(out["select_grasp_score"].cpu().numpy(), exp[p + "select_grasp_score"], rtol=0.0, atol=ATOL)
        assert out["select_grasp_class"].shape[0] == sum(b["keep3"]) > 0
    print("configs[2] B=8 with the refine stage: stage-2 grasps max abs err vs reference %.3e" % worst)
