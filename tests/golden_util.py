"""Shared helpers for the golden-fixture tests: regenerate the seeded inputs that
tests/golden/make_golden.py used and load the expected outputs."""
import hashlib
import json
import os

import numpy as np
import torch

from regnet_for_3d_grasping_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def meta():
    with open(os.path.join(GOLDEN, "golden_meta.json")) as f:
        return json.load(f)


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def scenes(cfg, device="cpu"):
    return synthetic.make_batch(cfg["scene_seed"], cfg["B"], cfg["N"], device=device)


def pseudo_scores(seed, B, N):
    rng = np.random.default_rng(seed)
    s = rng.uniform(0.0, 1.0, (B, N)).astype(np.float32)
    if B > 1:
        s[1] = (s[1] * 0.5).astype(np.float32)
        s[1, rng.choice(N, 40, replace=False)] = 0.9
    return torch.from_numpy(s)


def pseudo_feature(seed, B, N, F=256):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.normal(0.0, 1.0, (B, N, F)).astype(np.float32))


def build_scorenet(m, device="cpu"):
    from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
    net = ScoreNetwork(training=True)
    net.load_state_dict(synthetic.seeded_state_dict(net, m["cfg"]["score_weights_seed"]))
    bn = net.extrat_featurePN2.bn_score
    bn.running_mean.fill_(m["bn_score"]["running_mean"])
    bn.running_var.fill_(m["bn_score"]["running_var"])
    bn.weight.data.fill_(m["bn_score"]["weight"])
    bn.bias.data.fill_(m["bn_score"]["bias"])
    return net.to(device).eval()


def build_regionnet(m, device="cpu"):
    from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
    cfg = m["cfg"]
    net = GripperRegionNetwork(training=True, group_num=cfg["params"][2], gripper_num=cfg["gripper_num"],
                               grasp_score_threshold=cfg["grasp_score_threshold"], radius=cfg["gripper_params"][2],
                               reg_channel=cfg["reg_channel"])
    net.load_state_dict(synthetic.seeded_state_dict(net, cfg["region_weights_seed"]))
    return net.to(device).eval()


class OpRecorder:
    """Wraps the index-producing ops of an extension module and records their outputs in call order."""

    NAMES = ("farthest_point_sample", "ball_query", "point_search")

    def __init__(self, monkeypatch, ext):
        self.log = []
        for name in self.NAMES:
            orig = getattr(ext, name)

            def wrapped(*a, _orig=orig, _name=name):
                out = _orig(*a)
                outs = out if isinstance(out, (list, tuple)) else [out]
                self.log.append((_name, list(outs)))
                return out
            monkeypatch.setattr(ext, name, wrapped)

    def check_against(self, expected_ops):
        assert [n for n, _ in self.log] == [o["op"] for o in expected_ops]
        for (name, outs), exp in zip(self.log, expected_ops):
            assert list(outs[0].shape) == exp["shape"], name
            assert sha(outs[0]) == exp["index_sha256"], "%s index mismatch" % name
            if exp["aux_sha256"] is not None and name == "ball_query":
                assert sha(outs[1]) == exp["aux_sha256"], "%s count mismatch" % name


def loss_inputs(seed, n=96, A=4):
    """Seeded inputs for the stand-alone loss checks (shared with tests/test_golden_cpu.py)."""
    rng = np.random.default_rng(seed)
    f = lambda *s: torch.from_numpy(rng.normal(0, 1, s).astype(np.float32))
    centres = f(n, 3) * 0.2
    ground = torch.full((2, n // 2, 10), -1.0)
    has = torch.from_numpy(rng.uniform(0, 1, n) < 0.8)
    g = torch.cat([centres + f(n, 3) * 0.01, torch.nn.functional.normalize(f(n, 3), dim=1), f(n, 1) * 0.8,
                   torch.from_numpy(rng.uniform(0, 1, (n, 3)).astype(np.float32))], 1)
    g[:, 3:6] = torch.where(g[:, 3:4] < 0, -g[:, 3:6], g[:, 3:6])
    ground.view(-1, 10)[has] = g[has]
    stage2 = dict(first_grasp=f(n, A, 10) * 0.3, first_cls=f(n, A), centres=centres, ground=ground)
    m = 80
    next_gt = torch.cat([f(m, 3) * 0.2, torch.nn.functional.normalize(f(m, 3), dim=1), f(m, 1) * 0.8,
                         torch.from_numpy(rng.uniform(0, 1, (m, 3)).astype(np.float32))], 1)
    next_grasp = next_gt.clone()
    far = torch.from_numpy(rng.uniform(0, 1, m) < 0.5)
    next_grasp[far, :3] += 0.05
    next_grasp[:, :3] += f(m, 3) * 0.004
    next_grasp[:, 3:6] = torch.nn.functional.normalize(next_grasp[:, 3:6] + f(m, 3) * 0.2, dim=1)
    next_grasp[:, 6] += f(m) * 0.3
    refine = dict(next_grasp=next_grasp, next_x_cls=f(m, 2), next_x_reg=f(m, 10) * 0.1, next_gt=next_gt)
    return stage2, refine


def circulant_state(key, shape, dtype, seed=8):
    """numpy-seeded, O(1), WELL-CONDITIONED fill of one state_dict entry that still compresses (the reference-pickled
    checkpoint fixtures ckpt_*_8.model.gz: 28 MB of fp32 as a few hundred KB).  A weight matrix (O, I, ...) is block
    circulant: row o is ``g_b`` rotated by ``o mod I`` with ``g_b ~ N(0, 2 / I)`` drawn per block ``b = o // I`` from
    ``default_rng([seed, crc32(key), b])`` -- every output channel is a lag of the circular correlation of a random vector
    with the input, so the rows are as decorrelated as independent draws (unlike ckpt_*_7's 13-periodic ramp, whose rows
    are shifted copies of ONE ramp), while consecutive rows are byte-shifted copies of each other, which LZ77 finds.
    BatchNorm vectors and biases follow synthetic.seeded_state_dict's distributions.  ``key`` without the DataParallel
    ``module.`` prefix."""
    import zlib
    crc = zlib.crc32(key.encode())
    leaf = key.rsplit(".", 1)[-1]
    rng = np.random.default_rng([seed, crc])
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=dtype)
    if leaf == "running_mean":
        val = rng.normal(0.0, 0.1, shape)
    elif leaf == "running_var":
        val = rng.uniform(0.5, 1.5, shape)
    elif len(shape) >= 2:
        O, I = shape[0], int(np.prod(shape[1:]))
        rows = np.empty((O, I), np.float32)
        for b in range((O + I - 1) // I):
            g = (np.random.default_rng([seed, crc, b]).normal(0.0, 1.0, I) * np.sqrt(2.0 / I)).astype(np.float32)
            idx = (np.arange(I)[None, :] + np.arange(min(I, O - b * I))[:, None]) % I
            rows[b * I:b * I + idx.shape[0]] = g[idx]
        val = rows.reshape(shape)
    elif ".bn" in key or "bn_" in key or key.startswith("bn"):
        val = rng.uniform(0.8, 1.2, shape) if leaf == "weight" else rng.normal(0.0, 0.1, shape)
    else:
        val = rng.normal(0.0, 0.05, shape)
    return torch.from_numpy(np.asarray(val, dtype=np.float32)).to(dtype)


def meta_train():
    with open(os.path.join(GOLDEN, "s4_meta.json")) as f:
        return json.load(f)


# ---- stage S5: dataset records (tests/golden/make_golden_dataset.py, tests/test_scoredataset_cpu.py) ----------
# case -> (directory layout, split tag, data_seed, all_points_num)
DATASET_CASES = {
    "train": ("training", "train", 1, 256),
    "validate": ("training", "validate", 1, 256),
    "test": ("training", "test", 1, 512),
    "eval_train": ("eval", "train", 3, 192),
    "eval_heldout": ("eval", "validate", 3, 512),
}
DATASET_ITEMS = (0, 5, 11)


def dataset_records(tmp):
    """Writes the synthetic record trees and returns {'training': root, 'eval': root} (roots as handed to
    ScoreDataset).  90 + 7 + 90 records of 150..400 points; deterministic."""
    import os
    import numpy as np
    from regnet_for_3d_grasping_amd import scoredataset, synthetic
    rng = np.random.default_rng(515)
    roots = {"training": os.path.join(tmp, "data"), "eval": os.path.join(tmp, "eval_data")}
    layout = (("data/training_data", 90), ("data/training_data_test", 7), ("eval_data", 90))
    for sub, count in layout:
        os.makedirs(os.path.join(tmp, sub))
        for i in range(count):
            n = int(rng.integers(150, 400))
            scene = synthetic.make_scene(int(rng.integers(0, 1 << 30)), n)
            label = (scene[:, 2] > 0.7525).astype(np.float32) * rng.integers(1, 9, n)
            score = rng.uniform(0, 1.5, n) * (label > 0)
            scoredataset.write_record(os.path.join(tmp, sub, "scene_%04d.p" % i), scene, score, label,
                                      synthetic.make_grasp_labels(scene, i, every=5))
    return roots


# ---- view-cloud collision filter (tests/golden/make_golden_collision.py -> s6_collision.npz) -----------------------
def _collision_inputs(seed, n_points, n_grasps):
    """A synthetic table-top scene and grasps scattered on / above its surfaces (some collide, some hang in the air)."""
    from regnet_for_3d_grasping_amd import synthetic
    pts = synthetic.make_scene(seed, n_points)[:, :3].astype(np.float32)
    rng = np.random.default_rng(seed + 7)
    g = np.zeros((n_grasps, 8), dtype=np.float32)
    anchor = pts[rng.integers(0, n_points, n_grasps)]
    g[:, :3] = anchor + rng.normal(0, 0.012, (n_grasps, 3)).astype(np.float32)
    g[:, 2] += rng.uniform(0.0, 0.08, n_grasps).astype(np.float32)
    g[:, 3:6] = rng.normal(size=(n_grasps, 3)).astype(np.float32)
    g[:, 6] = rng.uniform(-1.2, 1.2, n_grasps).astype(np.float32)
    g[:, 7] = rng.uniform(0, 1, n_grasps).astype(np.float32)
    g[0, 3:6] = 0.0                      # zero axis -> the reference's fallbacks (:139, :144, :149)
    g[1, 3:6] = [0.0, 0.0, 1.0]          # axis_y || z -> zero axis_x
    return pts, g


COLLISION_CASES = [dict(seed=4100, n_points=6000, n_grasps=400, table_height=0.75, depth=0.06, width=0.08),
                   dict(seed=4200, n_points=12000, n_grasps=300, table_height=0.75, depth=0.05, width=0.06)]


def collision_case(i):
    c = COLLISION_CASES[i]
    return _collision_inputs(c["seed"], c["n_points"], c["n_grasps"])


# validation flavour (eval_validate): view cloud + a denser "scene" cloud with normals; grasps as (B,8) rows
VALIDATE_CASES = [dict(seed=4400, n_view=5000, n_scene=20000, n_grasps=300, view_num=1, table_height=0.75, depth=0.06, width=0.08),
                  dict(seed=4500, n_view=9000, n_scene=18000, n_grasps=250, view_num=3, table_height=0.75, depth=0.055, width=0.085)]


def validate_case(i):
    """-> (data dict with view_cloud / scene_cloud / scene_normal, grasps (B,8))."""
    from regnet_for_3d_grasping_amd import synthetic
    c = VALIDATE_CASES[i]
    scene = synthetic.make_scene(c["seed"], c["n_scene"])[:, :3].astype(np.float32)
    rng = np.random.default_rng(c["seed"] + 3)
    view = scene[rng.choice(c["n_scene"], c["n_view"], replace=False)]
    normal = rng.normal(size=(c["n_scene"], 3)).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    n = c["n_grasps"]
    g = np.zeros((n, 8), dtype=np.float32)
    # top-down grasps over object tops (fingers straddle the box when it is narrower than the opening) ...
    tops = view[view[:, 2] > 0.775]
    g[:, :3] = tops[rng.integers(0, len(tops), n)] + rng.normal(0, 0.004, (n, 3)).astype(np.float32)
    g[:, 2] += rng.uniform(0.01, 0.05, n).astype(np.float32)
    phi = rng.uniform(0, np.pi, n)
    g[:, 3], g[:, 4] = np.cos(phi), np.sin(phi)
    g[:, 6] = (-np.pi / 2 + rng.normal(0, 0.08, n)).astype(np.float32)   # approach = cos t * axis_x + sin t * z -> down
    # ... and a quarter of arbitrary ones
    k = n // 4
    g[:k, :3] = view[rng.integers(0, c["n_view"], k)] + rng.normal(0, 0.008, (k, 3)).astype(np.float32)
    g[:k, 3:6] = rng.normal(size=(k, 3)).astype(np.float32)
    g[:k, 6] = rng.uniform(-1.2, 1.2, k).astype(np.float32)
    g[:, 7] = rng.uniform(0, 1, n).astype(np.float32)
    return {"view_cloud": view, "scene_cloud": scene, "scene_normal": normal}, g


# ---- stage S7: full-size fixtures (tests/golden/make_golden_fullsize.py -> s7_*.npz, s7_meta.json) -----------------
def meta_full():
    with open(os.path.join(GOLDEN, "s7_meta.json")) as f:
        return json.load(f)


def build_scorenet_full(m, device="cpu"):
    """ScoreNet with the S7 weights and score-head calibration."""
    from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
    net = ScoreNetwork(training=True)
    net.load_state_dict(synthetic.seeded_state_dict(net, m["cfg"]["score_weights_seed"]))
    bn = net.extrat_featurePN2.bn_score
    bn.running_mean.fill_(m["bn_score"]["running_mean"])
    bn.running_var.fill_(m["bn_score"]["running_var"])
    bn.weight.data.fill_(m["bn_score"]["weight"])
    bn.bias.data.fill_(m["bn_score"]["bias"])
    return net.to(device).eval()


def check_ops_per_scene(log, expected_ops, scenes):
    """``log``: OpRecorder.log of a forward over the scenes ``scenes`` (indices into the fixture's batch); every index
    tensor must equal the reference's, scene by scene (SHA-256 of the per-scene slice)."""
    assert [n for n, _ in log] == [o["op"] for o in expected_ops]
    for (name, outs), exp in zip(log, expected_ops):
        assert list(outs[0].shape[1:]) == exp["shape"][1:], name
        for i, b in enumerate(scenes):
            assert sha(outs[0][i]) == exp["index_sha256"][b], "%s index mismatch (scene %d)" % (name, b)
            if exp["aux_sha256"] is not None and name == "ball_query":
                assert sha(outs[1][i]) == exp["aux_sha256"][b], "%s count mismatch (scene %d)" % (name, b)


def region_inputs_full(m, device="cpu"):
    """S7c inputs: scenes 0..B-1 of S7a, the REFERENCE's scores for them, the seeded pseudo feature map."""
    cfg = m["cfg"]
    Bc, N = cfg["c"]["B"], cfg["a"]["N"]
    pc = synthetic.make_batch(cfg["a"]["scene_seed"], cfg["a"]["B"], N)[:Bc].contiguous().to(device)
    score = torch.from_numpy(load("s7a_scorenet_25600.npz")["score"][:Bc].copy()).to(device)
    feat = pseudo_feature(cfg["c"]["feature_seed"], Bc, N).to(device)
    return pc, score, feat


def build_regionnet_full(m, device="cpu"):
    from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
    cfg = m["cfg"]
    net = GripperRegionNetwork(training=True, group_num=cfg["params"][2], gripper_num=cfg["gripper_num"],
                               grasp_score_threshold=cfg["grasp_score_threshold"], radius=cfg["gripper_params"][2],
                               reg_channel=cfg["reg_channel"])
    net.load_state_dict(synthetic.seeded_state_dict(net, cfg["region_weights_seed"]))
    return net.to(device).eval()


# ---- stage S9: configs[2] with the refine stage running (tests/golden/make_golden_refine.py -> s9_*) ----------------
def meta_refine():
    with open(os.path.join(GOLDEN, "s9_meta.json")) as f:
        return json.load(f)


def build_regionnet_refine(m7, m9, device="cpu"):
    """The S7 region network with S9's calibrated last BatchNorms (the constants the generator derived on the REFERENCE's
    module, loaded -- not re-derived)."""
    net = build_regionnet_full(m7, "cpu")
    synthetic.apply_region_calibration(net, m9["region_calibration"])
    return net.to(device).eval()


class CropSpy:
    """Records what ``get_gripper_region_transform`` returns inside ``GripperRegionNetwork.refine_forward`` and, when
    ``forced`` holds the reference's stage-2 grasps, teacher-forces the crop with them (as test_s3 does: fp32 noise in
    ``next_grasp`` must not move a point across a box face); the grasps the network itself decoded are kept in ``own``."""

    def __init__(self, monkeypatch, grn, forced=None):
        self.calls, self.own, self.forced = [], [], forced
        orig = grn.get_gripper_region_transform

        def spy(group_points, group_index, grasp, *a, **k):
            self.own.append(grasp.detach().clone())
            if self.forced is not None:
                want = self.forced[len(self.calls)]
                grasp = torch.as_tensor(want, dtype=grasp.dtype).to(grasp.device)
            out = orig(group_points, group_index, grasp, *a, **k)
            self.calls.append({"index_inall": out[2].detach().clone(), "valid": out[3].detach().clone()})
            return out
        monkeypatch.setattr(grn, "get_gripper_region_transform", spy)
