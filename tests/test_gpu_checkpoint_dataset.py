"""SURVEY.md section 8f ranks 1-2 on the GPU: a checkpoint pickled by the REFERENCE's own classes is restored with
``checkpoint.construct_*`` and run through the HIP path, and batches drawn by ``ScoreDataset`` from record files are
fed through the production pipeline -- both against the oracle-backed mirror on the CPU (same weights, same inputs),
floats within north_star's 1e-4 (relative to the tensor's scale: the fixture weights are not O(1))."""
import contextlib
import gzip
import io
import os
import shutil
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_util  # noqa: E402


def _unzip(name, tmp_path):
    dst = os.path.join(str(tmp_path), name[:-3].replace("ckpt_", ""))
    with gzip.open(os.path.join(HERE, "golden", name), "rb") as src, open(dst, "wb") as out:
        shutil.copyfileobj(src, out)
    return dst


def _rel_err(got, want):
    scale = float(want.abs().max()) or 1.0
    return float((got.double().cpu() - want.double()).abs().max()) / scale


def test_reference_checkpoints_run_on_the_hip_path(tmp_path):
    """utils.py:59-90 restore + forward.  The reference-pickled weights (tests/golden/make_golden_ckpt.py) are a
    deterministic +-6/16 pattern -- a weight scale quite unlike the seeded synthetic one, so this also probes the
    tolerance margin of the fp32-MFMA path away from the bench's weights."""
    from oracle.install import oracle_backend
    from regnet_for_3d_grasping_amd import checkpoint, pipeline, synthetic
    score_cpu, r0 = checkpoint.construct_scorenet(True, obj_class_num=2, model_path=_unzip("ckpt_score_7.model.gz", tmp_path),
                                                  map_location="cpu")
    region_cpu, r1 = checkpoint.construct_rnet(True, True, 256, 64, 0.5, 0.06, 10,
                                               model_path=_unzip("ckpt_region_7.model.gz", tmp_path), map_location="cpu")
    assert (r0, r1) == (8, 8)
    score_cpu.eval(); region_cpu.eval()
    pc = synthetic.make_batch(5100, 2, 6144)
    with oracle_backend():
        synthetic.calibrate_score_head(score_cpu, pc)      # the pattern leaves bn_score saturated: re-centre it (both sides)
        np.random.seed(5)
        want = pipeline.forward_scenes(score_cpu, region_cpu, pc)
    import copy
    score_gpu, region_gpu = copy.deepcopy(score_cpu).to(DEV).eval(), copy.deepcopy(region_cpu).to(DEV).eval()
    np.random.seed(5)
    got = pipeline.forward_scenes(score_gpu, region_gpu, pc.to(DEV))
    assert torch.isfinite(got["all_feature"]).all() and torch.isfinite(got["score"]).all()
    err_f, err_s = _rel_err(got["all_feature"], want["all_feature"]), _rel_err(got["score"], want["score"])
    # The pattern weights make the segmentation head ILL-CONDITIONED (every layer's rows are shifted copies of one
    # 13-periodic ramp, so conv_score's output is a small difference of large sums): measure how much the fp32 CPU head
    # itself moves when its input -- the 256-channel feature, which agrees to ~1e-5 -- is perturbed at that level, and
    # hold the GPU score to the stated 1e-4 plus that amplification.  (Measured: feature 1.0e-5, score 6.8e-3 against a
    # sensitivity of the same order; with the seeded O(1) weights of the other tests the score agrees to < 1e-4.)
    seg = score_cpu.extrat_featurePN2
    with torch.no_grad():
        F0 = want["all_feature"].transpose(1, 2).contiguous()
        noise = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, tuple(F0.shape)).astype(np.float32))
        def head(F):
            return torch.sigmoid(seg.bn_score(seg.conv_score(seg.mlp(F))))[:, 0, :]
        sens = float((head(F0 + noise * (max(err_f, 1e-6) * float(F0.abs().max()))) - head(F0)).abs().max())
    print("checkpoint forward: feature rel err %.2e, score abs err %.2e, head sensitivity at that input error %.2e"
          % (err_f, err_s, sens))
    assert err_f <= 1e-4
    assert err_s <= 1e-4 + 3.0 * sens
    # Region stage, teacher-forced from the CPU scores (the centre picker thresholds the score at 0.5: with this head a
    # 7e-3 score difference moves points across it): same numpy stream -> identical centres and groups; grasps within
    # tolerance relative to their scale.
    from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
    pc_gpu = pc.to(DEV)
    np.random.seed(5)
    with torch.no_grad():
        grp = get_grasp_allobj(pc_gpu, want["score"].to(DEV), pipeline.PARAMS, [])
        with contextlib.redirect_stdout(io.StringIO()):
            res = region_gpu(grp[3], grp[5], grp[2], grp[4], grp[0], grp[1], pc_gpu, got["all_feature"],
                             pipeline.GRIPPER_PARAMS, None, [])
    assert torch.equal(grp[1].cpu(), want["center_pc_index"])
    assert torch.equal(grp[2].cpu(), want["pc_group_index"])
    assert torch.equal(grp[4].cpu(), want["pc_group_more_index"])
    err_g = _rel_err(res[0], want["next_grasp"])
    print("checkpoint region stage: next_grasp rel err %.2e" % err_g)
    assert err_g <= 1e-4


def test_reference_seeded_checkpoints_hold_1e4_on_the_hip_path(tmp_path):
    """The second reference-pickled pair (tests/golden/ckpt_*_8.model.gz: written by the REFERENCE's classes with
    numpy-seeded O(1) block-circulant weights, tests/golden/make_golden_ckpt.py + golden_util.circulant_state) restored by
    utils.py:59-90's rules and run end to end: scores, features and grasps of the HIP path within north_star's 1e-4 of the
    oracle-backed CPU mirror -- absolute for the scores, NO sensitivity allowance (the +-6/16 pattern of ckpt_*_7 needs
    one: its head is ill-conditioned) -- and the region stage's indices identical without teacher forcing."""
    from oracle.install import oracle_backend
    from regnet_for_3d_grasping_amd import checkpoint, pipeline, synthetic
    score_cpu, r0 = checkpoint.construct_scorenet(True, obj_class_num=2, model_path=_unzip("ckpt_score_8.model.gz", tmp_path),
                                                  map_location="cpu")
    region_cpu, r1 = checkpoint.construct_rnet(True, True, 256, 64, 0.5, 0.06, 10,
                                               model_path=_unzip("ckpt_region_8.model.gz", tmp_path), map_location="cpu")
    assert (r0, r1) == (9, 9)
    score_cpu.eval(); region_cpu.eval()
    pc = synthetic.make_batch(5100, 2, 6144)
    with oracle_backend():
        synthetic.calibrate_score_head(score_cpu, pc)      # scores straddle the 0.5 threshold (both sides get the same BN)
        np.random.seed(5)
        want = pipeline.forward_scenes(score_cpu, region_cpu, pc)
    import copy
    score_gpu, region_gpu = copy.deepcopy(score_cpu).to(DEV).eval(), copy.deepcopy(region_cpu).to(DEV).eval()
    np.random.seed(5)
    got = pipeline.forward_scenes(score_gpu, region_gpu, pc.to(DEV))
    err_s = float((got["score"].cpu() - want["score"]).abs().max())
    err_f = float(((got["all_feature"].cpu() - want["all_feature"]).abs() / (1.0 + want["all_feature"].abs())).max())
    positives = [int(v) for v in (want["score"] > 0.5).sum(1)]
    print("seeded reference checkpoint: score abs err %.2e, feature err %.2e, positives %s" % (err_s, err_f, positives))
    assert min(positives) > 64 and max(positives) < 6144 - 64       # a real centre selection, not a fallback branch
    assert err_s <= 1e-4 and err_f <= 1e-4
    assert torch.equal(got["center_pc_index"].cpu(), want["center_pc_index"])
    assert torch.equal(got["pc_group_index"].cpu(), want["pc_group_index"])
    assert torch.equal(got["pc_group_more_index"].cpu(), want["pc_group_more_index"])
    assert got["next_grasp"].shape == want["next_grasp"].shape
    err_g = float((got["next_grasp"].cpu() - want["next_grasp"]).abs().max())
    print("seeded reference checkpoint: next_grasp abs err %.2e (|grasp| max %.2f)" % (err_g, float(want["next_grasp"].abs().max())))
    assert err_g <= 1e-4 * max(1.0, float(want["next_grasp"].abs().max()))


def test_dataset_batches_through_the_pipeline(tmp_path):
    """scoredataset.py:60-81 records -> DataLoader batches (resampled WITH replacement: the records hold a few hundred
    points, so every scene is full of duplicated points -- the FPS tie rules matter) -> ForwardPipeline."""
    from oracle.install import oracle_backend
    from regnet_for_3d_grasping_amd import pipeline, synthetic
    from dataset_utils.scoredataset import ScoreDataset      # the reference's import path
    roots = golden_util.dataset_records(str(tmp_path))
    N = 6144
    ds = ScoreDataset(N, roots["training"], "train", 1, [0.06, 0.08])
    np.random.seed(3)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0)
    batches = []
    for i, (view, score, label, path, width) in enumerate(loader):
        assert view.shape == (2, N, 6) and view.dtype == torch.float32
        batches.append(view)
        if i == 2:
            break
    score_cpu, region_cpu = pipeline.build_models("cpu")
    with oracle_backend():
        synthetic.calibrate_score_head(score_cpu, batches[0])
        np.random.seed(9)
        want = [pipeline.forward_scenes(score_cpu, region_cpu, b) for b in batches]
    score_gpu, region_gpu = pipeline.build_models(DEV)
    score_gpu.load_state_dict(score_cpu.state_dict())
    region_gpu.load_state_dict(region_cpu.state_dict())
    np.random.seed(9)
    pipe = pipeline.ForwardPipeline(score_gpu, region_gpu)
    got = list(pipe.run(iter([b.to(DEV) for b in batches])))
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert float((g["score"].cpu() - w["score"]).abs().max()) <= 1e-4
        assert _rel_err(g["all_feature"], w["all_feature"]) <= 1e-4
        for key in ("center_pc_index", "pc_group_index", "pc_group_more_index"):
            assert torch.equal(g[key].cpu(), w[key]), key


def test_dataset_item_on_the_gpu_equals_the_host_item(tmp_path):
    """ScoreDataset.gpu_item / gpu_batch (csrc/dataset.hip + the device-side numpy stream) against __getitem__
    (scoredataset.py:60-81) under the same seed: same picked rows (with AND without replacement, incl. a 30 000-point
    record whose shuffle is replayed in global memory), same colour gains (float64 products rounded to float32), tanh
    to 1e-6, same numpy stream position afterwards."""
    from regnet_for_3d_grasping_amd import np_random, region_ops, scoredataset, synthetic
    roots = golden_util.dataset_records(str(tmp_path))
    big = synthetic.make_scene(123, 30000)
    rng = np.random.default_rng(5)
    label = (big[:, 2] > 0.7525).astype(np.float32) * rng.integers(1, 9, 30000)
    scoredataset.write_record(os.path.join(roots["training"], "training_data", "zz_big_scene.p"), big,
                              rng.uniform(0, 1.5, 30000) * (label > 0), label, synthetic.make_grasp_labels(big, 1))
    for N in (256, 1024, 25600):        # 256: some records have more points (no replacement); 1024 / 25600: all fewer, except the big one
        ds = scoredataset.ScoreDataset(N, roots["training"], "train", 1, [0.06, 0.08])
        names = list(ds.data_name)
        items = [0, 5, 11] + ([names.index("zz_big_scene.p")] if "zz_big_scene.p" in names else [])
        np.random.seed(77)
        want = [ds[i] for i in items]
        after = int(np.random.randint(0, 2 ** 31 - 1))
        np.random.seed(77)
        pc, score, label_g, paths, widths = ds.gpu_batch(items, DEV)
        torch.cuda.synchronize()
        assert int(np.random.randint(0, 2 ** 31 - 1)) == after, "numpy stream position, N=%d" % N
        region_ops.raise_if_out_of_range()
        assert pc.shape == (len(items), N, 6) and widths.shape == (len(items), 2)
        for k, w in enumerate(want):
            np.testing.assert_array_equal(pc[k].cpu().numpy(), w[0].astype(np.float32))
            np.testing.assert_allclose(score[k].cpu().numpy(), w[1], rtol=0, atol=1e-6)
            np.testing.assert_array_equal(label_g[k].cpu().numpy(), w[2])
            assert paths[k] == w[3]


def test_model_saved_after_a_fused_forward_holds_no_package_objects(tmp_path):
    """checkpoint.save_model of a network that has already run the fused forward (packed-weight caches and signature lists
    hang on its modules then): the pickle still names only the reference's class paths, the caches survive the save, and the
    file is not inflated by a second copy of the weights."""
    import pickletools
    import zipfile
    from regnet_for_3d_grasping_amd import checkpoint, pipeline, synthetic
    score_net, region_net = pipeline.build_models(DEV)
    pc = synthetic.make_batch(4100, 1, 6144, device=DEV)
    np.random.seed(4)
    want = pipeline.forward_scenes(score_net, region_net, pc)
    assert any(k.startswith("_regnet_") for m in score_net.modules() for k in m.__dict__)
    sizes = []
    for net, name in ((score_net, "score_0.model"), (region_net, "region_0.model")):
        path = os.path.join(str(tmp_path), name)
        checkpoint.save_model(net, path)
        with zipfile.ZipFile(path) as z:
            data = z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])
        strings = [arg for op, arg, _ in pickletools.genops(data) if isinstance(arg, str)]
        assert not any("regnet_for_3d_grasping_amd" in s for s in strings)
        sizes.append((os.path.getsize(path), sum(v.numel() * v.element_size() for v in net.state_dict().values())))
    for on_disk, state in sizes:
        assert on_disk < 1.2 * state + 1_000_000
    assert any(k.startswith("_regnet_") for m in score_net.modules() for k in m.__dict__)     # caches restored
    np.random.seed(4)
    got = pipeline.forward_scenes(score_net, region_net, pc)
    assert torch.equal(got["score"], want["score"]) and torch.equal(got["next_grasp"], want["next_grasp"])
