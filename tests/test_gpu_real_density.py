"""The density-matched scenes (``synthetic.make_scene(density="real")``: level-1 neighbourhood sizes like the reference's own
clouds, tests/golden/real_density_hist.json) through the HIP path at configs[2]'s full size, 8 x 25 600 points, against the
CPU oracle: sampling, ball query and member counts of every level bit-exact for all 8 scenes; the whole forward of two of
them (ScoreNet + region grouping + grasp heads) against the oracle-backed mirror within north_star's 1e-4.  These scenes
put most level-1 neighbourhoods in the classes the uniform scenes have few of (full, and <= 32 members), i.e. other
branches of ``sa_chain_kernel``'s tile skipping and pairing."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


def _sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def test_real_density_geometry_is_bit_exact_at_8x25600():
    from oracle import pn2_ext_oracle as oracle
    from regnet_for_3d_grasping_amd import pn2_ext, synthetic
    with open(os.path.join(HERE, "golden", "real_density_hist.json")) as f:
        target = json.load(f)["mean_of_files"]
    B, N = 8, 25600
    pc = synthetic.make_batch(1000, B, N, density="real")
    xyz_c = pc.permute(0, 2, 1)[:, :3, :].contiguous()
    xyz_g = xyz_c.to(DEV)
    level_counts = []
    for M, radius in ((5120, 0.02), (1024, 0.08), (256, 0.32)):
        ctr_c = oracle.farthest_point_sample(xyz_c, M)
        ctr_g = pn2_ext.farthest_point_sample(xyz_g, M)
        assert torch.equal(ctr_g.cpu(), ctr_c), "FPS level with %d centroids" % M
        cx_c = torch.gather(xyz_c, 2, ctr_c[:, None, :].expand(B, 3, M)).contiguous()
        cx_g = cx_c.to(DEV)
        nbr_c, cnt_c = oracle.ball_query(xyz_c, cx_c, radius, 64)
        nbr_g, cnt_g = pn2_ext.ball_query(xyz_g, cx_g, radius, 64)
        assert torch.equal(cnt_g.cpu(), cnt_c) and torch.equal(nbr_g.cpu(), nbr_c), "ball query r = %g" % radius
        idx_c, d_c = oracle.point_search(xyz_c, cx_c, 3)
        idx_g, d_g = pn2_ext.point_search(xyz_g, cx_g, 3)
        assert torch.equal(idx_g.cpu(), idx_c) and torch.equal(d_g.cpu(), d_c), "3-NN onto %d keys" % M
        level_counts.append(cnt_c)
        xyz_c, xyz_g = cx_c, cx_g
    pn2_ext.raise_if_fps_failed()
    c = level_counts[0].float()
    got = {"mean": float(c.mean()), "le32": float((c <= 32).float().mean()), "le48": float((c <= 48).float().mean()),
           "eq64": float((c == 64).float().mean())}
    print("level-1 neighbourhoods of the density-matched scenes:", {k: round(v, 4) for k, v in got.items()}, "target", target)
    assert abs(got["mean"] - target["mean"]) <= 1.5
    for key in ("le32", "le48", "eq64"):
        assert abs(got[key] - target[key]) <= 0.03, (key, got[key], target[key])


def test_real_density_forward_matches_the_cpu_mirror():
    from oracle.install import oracle_backend
    from regnet_for_3d_grasping_amd import pipeline, synthetic
    B, N = 2, 25600
    pc = synthetic.make_batch(1004, B, N, density="real")
    score_cpu, region_cpu = pipeline.build_models("cpu")
    with oracle_backend():
        synthetic.calibrate_score_head(score_cpu, pc)
        np.random.seed(31)
        want = pipeline.forward_scenes(score_cpu, region_cpu, pc)
    score_gpu, region_gpu = pipeline.build_models(DEV)
    score_gpu.load_state_dict(score_cpu.state_dict())
    region_gpu.load_state_dict(region_cpu.state_dict())
    np.random.seed(31)
    got = pipeline.forward_scenes(score_gpu, region_gpu, pc.to(DEV))
    torch.cuda.synchronize()
    err_s = float((got["score"].cpu() - want["score"]).abs().max())
    err_f = float(((got["all_feature"].cpu() - want["all_feature"]).abs() / (1.0 + want["all_feature"].abs())).max())
    print("density-matched scenes, 2 x 25 600: score |err| %.2e, feature rel err %.2e, positives %s" % (
        err_s, err_f, [int(v) for v in (want["score"] > 0.5).sum(1)]))
    assert err_s <= 1e-4 and err_f <= 1e-4
    for key in ("center_pc_index", "pc_group_index", "pc_group_more_index"):
        assert _sha(got[key]) == _sha(want[key]), key
    assert got["next_grasp"].shape == want["next_grasp"].shape
    assert float((got["next_grasp"].cpu() - want["next_grasp"]).abs().max()) <= 1e-4


def test_real_density_pipeline_equals_sequential_at_8x25600():
    """The overlapped pipeline on density-matched batches of the bench's size returns the bits of the sequential forward."""
    from regnet_for_3d_grasping_amd import pipeline, synthetic
    score_net, region_net = pipeline.build_models(DEV)
    batches = [synthetic.make_batch(1000 + 8 * i, 8, 25600, device=DEV, density="real") for i in range(3)]
    synthetic.calibrate_score_head(score_net, batches[0])
    np.random.seed(9)
    want = [pipeline.forward_scenes(score_net, region_net, pc) for pc in batches]
    torch.cuda.synchronize()
    np.random.seed(9)
    got = list(pipeline.ForwardPipeline(score_net, region_net).run(iter(batches)))
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        for key in ("score", "all_feature", "center_pc_index", "pc_group_index", "pc_group_more_index", "next_grasp"):
            assert torch.equal(g[key], w[key]), key
