"""Seeded synthetic inputs for parity tests and bench.py.

Nothing here depends on the reference: scenes are drawn from ``numpy.random.default_rng`` and
weights from a per-key numpy generator, so the same bytes can be regenerated on the GPU box.
"""
import zlib

import numpy as np
import torch


def make_scene(seed, num_points=25600):
    """One table-top scene as float32 ``(num_points, 6)`` = xyz + rgb, order-shuffled.

    60 % of the points lie on a table plane (z = 0.75 m with 1 mm noise); 40 % on the visible
    faces (top, +x side, +y side) of eight boxes standing on it.  Point order is a random
    permutation -- ball-query early exit depends on it.
    """
    rng = np.random.default_rng(seed)
    n_table = int(round(num_points * 0.6))
    n_obj = num_points - n_table
    table = np.empty((n_table, 3), np.float64)
    table[:, 0] = rng.uniform(-0.40, 0.40, n_table)
    table[:, 1] = rng.uniform(-0.35, 0.35, n_table)
    table[:, 2] = 0.75 + rng.normal(0.0, 0.001, n_table)

    n_boxes = 8
    centre = rng.uniform([-0.30, -0.25], [0.30, 0.25], (n_boxes, 2))
    half = rng.uniform(0.02, 0.06, (n_boxes, 3))
    box = rng.integers(0, n_boxes, n_obj)
    face = rng.integers(0, 3, n_obj)  # 0 top, 1 +x side, 2 +y side
    u = rng.uniform(-1.0, 1.0, n_obj)
    v = rng.uniform(-1.0, 1.0, n_obj)
    hx, hy, hz = half[box, 0], half[box, 1], half[box, 2]
    obj = np.empty((n_obj, 3), np.float64)
    top, sx, sy = face == 0, face == 1, face == 2
    obj[:, 0] = centre[box, 0] + np.where(sx, hx, u * hx)
    obj[:, 1] = centre[box, 1] + np.where(sy, hy, np.where(sx, u * hy, v * hy))
    obj[:, 2] = 0.75 + np.where(top, 2.0 * hz, (v + 1.0) * hz)

    xyz = np.concatenate([table, obj], 0)
    rgb = rng.uniform(0.0, 1.0, (num_points, 3))
    scene = np.concatenate([xyz, rgb], 1).astype(np.float32)
    return scene[rng.permutation(num_points)]


def make_batch(first_seed, batch, num_points=25600, device="cpu"):
    """``(batch, num_points, 6)`` float32; scene ``i`` uses seed ``first_seed + i``."""
    arr = np.stack([make_scene(first_seed + i, num_points) for i in range(batch)], 0)
    return torch.from_numpy(arr).to(device)


def seeded_state_dict(module, seed):
    """Deterministic, torch-RNG-independent weights for ``module`` (keyed on parameter names).

    Conv/linear weights ~ N(0, 1/fan_in)·gain, biases small; BatchNorm affine and running
    statistics are randomised so that eval-mode activations are not degenerate (with default
    init every ScoreNet eval score sits at ≈0.48 and no point passes the 0.5 threshold).
    """
    out = {}
    for key, ref in module.state_dict().items():
        rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
        shape = tuple(ref.shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            val = np.zeros(shape, np.int64)
        elif leaf == "running_mean":
            val = rng.normal(0.0, 0.1, shape)
        elif leaf == "running_var":
            val = rng.uniform(0.5, 1.5, shape)
        elif ref.dim() >= 2:  # conv / linear weight
            fan_in = int(np.prod(shape[1:]))
            val = rng.normal(0.0, 1.0, shape) * np.sqrt(2.0 / fan_in)
        elif ".bn" in key or "bn_" in key or key.startswith("bn"):
            val = rng.uniform(0.8, 1.2, shape) if leaf == "weight" else rng.normal(0.0, 0.1, shape)
        else:  # conv / linear bias
            val = rng.normal(0.0, 0.05, shape)
        out[key] = torch.from_numpy(np.asarray(val)).to(ref.dtype)
    return out


def calibrate_score_head(score_net, pc):
    """Set ``bn_score``'s running statistics to those of ``conv_score``'s output on ``pc`` (and its
    affine to 2/0) so that eval-mode scores straddle the 0.5 centre-selection threshold instead of
    collapsing on one side.  Returns (mean, var)."""
    seg = score_net.extrat_featurePN2
    grabbed = {}
    hook = seg.conv_score.register_forward_hook(lambda m, i, o: grabbed.__setitem__("x", o.detach()))
    from . import fused
    was = fused.ENABLED
    fused.ENABLED = False  # the hook needs the module-granular head
    try:
        with torch.no_grad():
            score_net(pc)
    finally:
        fused.ENABLED = was
        hook.remove()
    x = grabbed["x"]
    mean, var = float(x.mean()), float(x.var(unbiased=False))
    with torch.no_grad():
        seg.bn_score.running_mean.fill_(mean)
        seg.bn_score.running_var.fill_(var)
        seg.bn_score.weight.fill_(2.0)
        seg.bn_score.bias.fill_(0.0)
    return mean, var


def make_grasp_labels(scene, seed, every=20):
    """Synthetic ground-truth grasps for a ``make_scene`` cloud, in the dataset's pickle layout
    (get_regiondataset.py:66-71): ``frame`` (G,4,4) with columns [approach | closing axis | normal |
    contact point] and ``antipodal_score`` (G,).  One grasp at every ``every``-th point above the table
    plane (z > 0.7525): the approach points mostly downwards with a random tilt, the closing axis is a
    random direction orthogonal to it."""
    rng = np.random.default_rng([seed, 77])
    xyz = np.asarray(scene)[:, :3].astype(np.float64)
    obj = np.nonzero(xyz[:, 2] > 0.7525)[0][::every]
    G = len(obj)
    approach = np.stack([rng.normal(0, 0.3, G), rng.normal(0, 0.3, G), -np.ones(G)], 1)
    approach /= np.linalg.norm(approach, axis=1, keepdims=True)
    rand = rng.normal(size=(G, 3))
    axis_y = rand - (rand * approach).sum(1, keepdims=True) * approach
    axis_y /= np.linalg.norm(axis_y, axis=1, keepdims=True)
    axis_z = np.cross(approach, axis_y)
    frame = np.zeros((G, 4, 4), np.float64)
    frame[:, :3, 0], frame[:, :3, 1], frame[:, :3, 2], frame[:, :3, 3] = approach, axis_y, axis_z, xyz[obj]
    frame[:, 3, 3] = 1.0
    return {"frame": frame.astype(np.float32), "antipodal_score": rng.uniform(0, 1, G).astype(np.float32)}
