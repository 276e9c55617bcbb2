"""Seeded synthetic inputs for parity tests and bench.py.

Nothing here depends on the reference: scenes are drawn from ``numpy.random.default_rng`` and
weights from a per-key numpy generator, so the same bytes can be regenerated on the GPU box.
"""
import zlib

import numpy as np
import torch


# density="real": relative areal point densities of four bands of the table (and of the boxes standing in them) and the
# share of the table's length each band takes -- fitted with scripts/real_density_fit.py so that the level-1 neighbourhood
# sizes (r = 0.02, K = 64, 5 120 FPS centroids) reproduce the reference clouds' histogram (tests/golden/real_density_hist.json,
# derived by scripts/real_density_hist.py from test_file/*_predict/*.p in the authoring container): mean 49.5 members,
# 23 % of the neighbourhoods with <= 32, 43 % with <= 48, 49 % full
REAL_DENSITY_BANDS = ((0.245, 18.5), (0.26, 31.0), (0.08, 44.0), (0.415, 71.0))
REAL_DENSITY_TABLE = (0.34, 0.30)      # half extents of the table of the density="real" scenes (m)


def make_scene(seed, num_points=25600, density="uniform"):
    """One table-top scene as float32 ``(num_points, 6)`` = xyz + rgb, order-shuffled.

    60 % of the points lie on a table plane (z = 0.75 m with 1 mm noise); 40 % on the visible
    faces (top, +x side, +y side) of eight boxes standing on it.  Point order is a random
    permutation -- ball-query early exit depends on it.

    ``density="uniform"`` (every fixture and the bench headline): table points uniform over the table.
    ``density="real"``: the same geometry on a smaller table whose areal point density varies in bands along x the way a
    depth camera's does with range (``REAL_DENSITY_BANDS``), matched to the reference clouds' level-1 neighbourhood-size
    histogram -- what decides how much work ``sa_chain_kernel`` skips (SURVEY.md 8d "real-density variant").
    """
    if density == "real":
        return _make_scene_real(seed, num_points)
    if density != "uniform":
        raise ValueError("density must be 'uniform' or 'real'")
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


def _make_scene_real(seed, num_points):
    rng = np.random.default_rng([seed, 4])
    hx_t, hy_t = REAL_DENSITY_TABLE
    share = np.array([b[0] for b in REAL_DENSITY_BANDS])
    dens = np.array([b[1] for b in REAL_DENSITY_BANDS])
    edges = -hx_t + 2.0 * hx_t * np.concatenate([[0.0], np.cumsum(share)])          # band i = [edges[i], edges[i + 1]) in x
    n_boxes = 8
    centre = rng.uniform([-hx_t + 0.06, -hy_t + 0.06], [hx_t - 0.06, hy_t - 0.06], (n_boxes, 2))
    half = rng.uniform(0.02, 0.06, (n_boxes, 3))
    box_band = np.clip(np.searchsorted(edges, centre[:, 0], side="right") - 1, 0, len(dens) - 1)
    # expected points of every surface piece = area x relative density, scaled to num_points
    table_area = share * (2.0 * hx_t) * (2.0 * hy_t)
    face_area = np.stack([4.0 * half[:, 0] * half[:, 1], 4.0 * half[:, 1] * half[:, 2], 4.0 * half[:, 0] * half[:, 2]], 1)
    weight = np.concatenate([table_area * dens, (face_area * dens[box_band][:, None]).reshape(-1)])
    piece = rng.choice(len(weight), num_points, p=weight / weight.sum())
    u = rng.uniform(-1.0, 1.0, num_points)
    v = rng.uniform(-1.0, 1.0, num_points)
    xyz = np.empty((num_points, 3), np.float64)
    nb = len(dens)
    on_table = piece < nb
    band = np.where(on_table, piece, 0)
    xyz[:, 0] = np.where(on_table, edges[band] + (u + 1.0) * 0.5 * (edges[band + 1] - edges[band]), 0.0)
    xyz[:, 1] = np.where(on_table, v * hy_t, 0.0)
    xyz[:, 2] = np.where(on_table, 0.75 + rng.normal(0.0, 0.001, num_points), 0.0)
    k = np.where(on_table, 0, piece - nb)
    box, face = k // 3, k % 3                                                          # 0 top, 1 +x side, 2 +y side
    hx, hy, hz = half[box, 0], half[box, 1], half[box, 2]
    top, sx, sy = face == 0, face == 1, face == 2
    ox = centre[box, 0] + np.where(sx, hx, u * hx)
    oy = centre[box, 1] + np.where(sy, hy, np.where(sx, u * hy, v * hy))
    oz = 0.75 + np.where(top, 2.0 * hz, (v + 1.0) * hz)
    obj = ~on_table
    xyz[obj, 0], xyz[obj, 1], xyz[obj, 2] = ox[obj], oy[obj], oz[obj]
    rgb = rng.uniform(0.0, 1.0, (num_points, 3))
    scene = np.concatenate([xyz, rgb], 1).astype(np.float32)
    return scene[rng.permutation(num_points)]


def make_batch(first_seed, batch, num_points=25600, device="cpu", density="uniform"):
    """``(batch, num_points, 6)`` float32; scene ``i`` uses seed ``first_seed + i``."""
    arr = np.stack([make_scene(first_seed + i, num_points, density) for i in range(batch)], 0)
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
    from . import fused
    if (pc.is_cuda and fused.ENABLED and not score_net.training and seg.k_score == 1
            and fused.supports_rowchain(seg, seg.fp_modules[-1])):
        # on the GPU: the fused forward's point feature (the last propagation block's output) through the head's four layers
        # on the native layer kernel; conv_score's output is one matrix-vector product away -- no module-granular forward of
        # the whole network (a cuDNN / MIOpen / rocBLAS pass over 8 x 25 600 points just to read one activation)
        with torch.no_grad():
            feat, _, _ = score_net(pc)                                   # (B, N, 256)
            B, N, C = feat.shape
            h = feat.reshape(B * N, C)
            h = h if h.is_contiguous() else h.contiguous()
            for layer in fused._packed_stack(seg.mlp, seg.mlp):
                h = fused.mlp_layer(h, layer.K, layer, B * N)
            w = seg.conv_score.weight.detach().reshape(-1).float()
            x = torch.mv(h, w)
            if seg.conv_score.bias is not None:
                x = x + seg.conv_score.bias.detach().float()[0]
            mean, var = float(x.mean()), float(x.var(unbiased=False))
            seg.bn_score.running_mean.fill_(mean)
            seg.bn_score.running_var.fill_(var)
            seg.bn_score.weight.fill_(2.0)
            seg.bn_score.bias.fill_(0.0)
        return mean, var
    grabbed = {}
    hook = seg.conv_score.register_forward_hook(lambda m, i, o: grabbed.__setitem__("x", o.detach()))
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


def _bn_input(bn, y):
    """Invert an eval-mode BatchNorm: its output ``y`` (rows, channels) -> the convolution output it was fed."""
    std = torch.sqrt(bn.running_var + bn.eps)
    return (y - bn.bias) / bn.weight * std + bn.running_mean


def _standardise(bn, x, weight, bias, channels=None):
    """Running statistics := those of ``x`` (rows, channels), affine := (weight, bias), on ``channels`` (default all)."""
    sel = slice(None) if channels is None else channels
    with torch.no_grad():
        bn.running_mean[sel] = x.mean(0)[sel]
        bn.running_var[sel] = x.var(0, unbiased=False)[sel]
        bn.weight[sel] = torch.as_tensor(weight, dtype=bn.weight.dtype, device=bn.weight.device).expand_as(bn.weight)[sel]
        bn.bias[sel] = torch.as_tensor(bias, dtype=bn.bias.dtype, device=bn.bias.device).expand_as(bn.bias)[sel]


REGION_CALIBRATION_KEYS = tuple("%s.%s.%s" % (net, bn, leaf)
                                for net, bn in (("extrat_feature_region", "bn_cls4"), ("extrat_feature_region", "bn_reg4"),
                                                ("extrat_feature_refine", "bn_formal_cls3"),
                                                ("extrat_feature_refine", "bn_formal_reg3"))
                                for leaf in ("weight", "bias", "running_mean", "running_var"))


def _stage2_affine(region_net, lift, spread, dev):
    """(weight, bias) of ``bn_reg4`` per (anchor, channel), flattened: see ``calibrate_region_head``."""
    A, C = region_net.anchor_number, region_net.reg_channel
    tmpl = region_net.templates.detach().float().to(dev).view(A, 4)
    w = torch.zeros(A, C, device=dev)
    b = torch.zeros(A, C, device=dev)
    w[:, 0:3], w[:, 3:6], w[:, 6] = spread[0], spread[1], spread[2]
    b[:, 2] = lift / float(region_net.radius)
    b[:, 3:6] = torch.tensor([1.0, 0.0, 0.0], device=dev).view(1, 3) - tmpl[:, :3]
    b[:, 6] = -0.5 - tmpl[:, 3]
    return w.view(-1), b.view(-1)


def set_region_head_affine(region_net, lift=0.008, spread=(0.12, 0.15, 0.05), refine_spread=0.05):
    """The TRAINING-mode counterpart of ``calibrate_region_head``: train-mode BatchNorm standardises with the batch's own
    statistics, so writing the affine of the four last BatchNorms is all it takes for the decoded stage-2 grasps to hold
    points in their closing boxes (refine losses on real rows instead of the ``len(gripper_mask) < 2`` skip).  No forward
    pass; running statistics untouched."""
    head, refine = region_net.extrat_feature_region, region_net.extrat_feature_refine
    A, C = region_net.anchor_number, region_net.reg_channel
    dev = head.bn_reg4.weight.device
    w, b = _stage2_affine(region_net, lift, spread, dev)
    reg_ch = (torch.arange(A * C, device=dev) % C) < 7
    with torch.no_grad():
        head.bn_cls4.weight.fill_(1.0)
        head.bn_cls4.bias.fill_(0.0)
        head.bn_reg4.weight[reg_ch] = w[reg_ch]
        head.bn_reg4.bias[reg_ch] = b[reg_ch]
        refine.bn_formal_cls3.weight.fill_(1.0)
        refine.bn_formal_cls3.bias.fill_(0.0)
        refine.bn_formal_reg3.weight.fill_(refine_spread)
        refine.bn_formal_reg3.bias.fill_(0.0)
    return region_net


def calibrate_region_head(region_net, run, lift=0.008, spread=(0.12, 0.15, 0.05), refine_spread=0.05):
    """Make a seeded ``GripperRegionNetwork`` decode to grasps whose closing box actually holds points, so that the
    refine stage runs (gripper_region_network.py:333: ``if len(gripper_mask) >= 2``) and its class head keeps a share
    of them.  ``run()`` pushes the calibration batch through ``region_net`` (called twice).

    With purely random weights the decoded grasps point anywhere, no crop has more than 5 points and the third network
    of BASELINE.json's configs[2] is a no-op.  As ``calibrate_score_head`` does for the scores, only the LAST BatchNorm
    of each branch is rewritten -- running statistics := the statistics of its input on the calibration batch (read
    back through the module outputs, so it works on the reference's modules and on the fused heads alike), affine:
      * anchor class (``bn_cls4``, pointnet2.py:148,181): (1, 0) -> all four anchors get picked;
      * stage-2 regression (``bn_reg4``, :156,185), per anchor ``a``: centre (ch 0-2) ~ spread[0] around
        (0, 0, lift / radius) -> the decoded centre floats ~``lift`` metres above its centre point ("up" is +z; a share
        sinks below the surface and stays invalid, so both branches of :532-544 stay exercised); closing axis (ch 3-5)
        ~ spread[1] around (1,0,0) - template_a -> horizontal; theta (ch 6) ~ spread[2] around -0.5 -> theta = -pi/2 ->
        approach = -z (:466-493); the score channels (7+) keep their seeded affine;
      * refine class (``bn_formal_cls3``, :214,244): (1, 0) -> both classes occur; refine regression
        (``bn_formal_reg3``, :219,248): (refine_spread, 0).
    Returns {state_dict key: tensor} of the sixteen tensors it wrote (``REGION_CALIBRATION_KEYS``) -- fixtures store them
    (tests/golden/make_golden_refine.py) and tests load them instead of re-deriving them from fp32 statistics."""
    head, refine = region_net.extrat_feature_region, region_net.extrat_feature_refine
    A, C = region_net.anchor_number, region_net.reg_channel
    grabbed = {}
    h1 = head.register_forward_hook(lambda m, i, o: grabbed.__setitem__("region", [t.detach() for t in o[:2]]))
    h2 = refine.register_forward_hook(lambda m, i, o: grabbed.__setitem__("refine", [t.detach() for t in o[:2]]))
    try:
        run()
        x_cls, x_reg = grabbed["region"]
        dev = x_cls.device
        _standardise(head.bn_cls4, _bn_input(head.bn_cls4, x_cls), 1.0, 0.0)
        # channels 7+ went through a sigmoid on their way out and keep their seeded affine
        reg_ch = (torch.arange(A * C, device=dev) % C) < 7
        x_reg = x_reg.reshape(-1, A * C)
        y = torch.where(reg_ch.view(1, -1), x_reg, torch.zeros_like(x_reg))
        w, b = _stage2_affine(region_net, lift, spread, dev)
        _standardise(head.bn_reg4, _bn_input(head.bn_reg4, y), w, b, channels=reg_ch)
        grabbed.pop("refine", None)
        run()
        if "refine" not in grabbed:
            raise RuntimeError("calibrate_region_head: fewer than 2 valid crops on the calibration batch")
        r_cls, r_reg = grabbed["refine"]
        _standardise(refine.bn_formal_cls3, _bn_input(refine.bn_formal_cls3, r_cls), 1.0, 0.0)
        _standardise(refine.bn_formal_reg3, _bn_input(refine.bn_formal_reg3, r_reg), refine_spread, 0.0)
    finally:
        h1.remove()
        h2.remove()
    state = region_net.state_dict()
    return {k: state[k].detach().clone() for k in REGION_CALIBRATION_KEYS}


def apply_region_calibration(region_net, constants):
    """Load what ``calibrate_region_head`` returned (tensors or nested lists, e.g. from a fixture's JSON)."""
    state = region_net.state_dict()
    with torch.no_grad():
        for k in REGION_CALIBRATION_KEYS:
            state[k].copy_(torch.as_tensor(constants[k], dtype=state[k].dtype).to(state[k].device))
    return region_net
