"""Python binding of the region-grouping kernels (csrc/region.hip): radius candidates, gripper
closing-box candidates and the fused feature gather + max-pool.  GPU tensors only."""
import numpy as np
import torch

from . import _lib
from .pn2_ext import _need_f32, _need_i64, _stream

_check = _lib.check
_L = _lib.lib


def sqrt_le_threshold(radius):
    """Largest float32 ``T`` such that ``float32(sqrt(T)) <= float32(radius)``.

    The reference tests ``torch.sqrt(d2) <= R`` (get_regiondataset.py:293-294).  Correctly
    rounded sqrt is monotonic, so that set equals ``{d2 <= T}``; comparing squared distances with
    ``T`` gives the same members without depending on the device's sqrt rounding."""
    r = np.float32(radius)
    if not r >= 0:
        return float("-inf")
    t = np.float32(np.float64(r) * np.float64(r))
    inf = np.float32(np.inf)
    while np.sqrt(t) > r:
        t = np.nextafter(t, -inf, dtype=np.float32)
    while np.sqrt(np.nextafter(t, inf, dtype=np.float32)) <= r:
        t = np.nextafter(t, inf, dtype=np.float32)
    return float(t)


def radius_candidates(pc, centres, radius):
    """pc (B,N,C>=3), centres (B,Nc,C'>=3), radius (float, already float32-rounded) ->
    cand (B,Nc,N) int32 ascending member indices (first ``count`` entries valid), count (B,Nc) int32."""
    _need_f32(pc, "pc")
    _need_f32(centres, "centres")
    if pc.stride(2) != 1 or centres.stride(2) != 1:
        pc, centres = pc.contiguous(), centres.contiguous()
    B, N, _ = pc.shape
    Nc = centres.shape[1]
    with torch.cuda.device(pc.device):
        cand = torch.empty((B, Nc, max(N, 1)), dtype=torch.int32, device=pc.device)
        count = torch.empty((B, Nc), dtype=torch.int32, device=pc.device)
        _check(_L.regnet_radius_group_f32(pc.data_ptr(), pc.stride(0), pc.stride(1), centres.data_ptr(),
                                          centres.stride(0), centres.stride(1), B, N, Nc,
                                          sqrt_le_threshold(radius), cand.size(2), cand.data_ptr(), count.data_ptr(),
                                          _stream(pc)), "radius_group")
    return cand, count


def select_positive(pc, score, threshold):
    """pc (B,N,C>=3), score (B,N) float32, threshold -> index (B,N) int64 (ascending ids of the points
    with score > threshold in the first ``count[b]`` slots), xyz (B,3,N) (their coordinates, padded with
    the first positive), count (B) int32."""
    _need_f32(pc, "pc")
    _need_f32(score, "score")
    if pc.stride(2) != 1:
        pc = pc.contiguous()
    if score.stride(1) != 1:
        score = score.contiguous()
    B, N, _ = pc.shape
    with torch.cuda.device(pc.device):
        index = torch.empty((B, max(N, 1)), dtype=torch.int64, device=pc.device)
        xyz = torch.empty((B, 3, max(N, 1)), dtype=torch.float32, device=pc.device)
        count = torch.empty((B,), dtype=torch.int32, device=pc.device)
        _check(_L.regnet_select_positive_f32(pc.data_ptr(), pc.stride(0), pc.stride(1), score.data_ptr(),
                                             score.stride(0), B, N, float(threshold), index.data_ptr(),
                                             xyz.data_ptr(), count.data_ptr(), _stream(pc)), "select_positive")
    return index, xyz, count


def resample_groups(pc, cand, pos):
    """pc (B,N,C) float32, cand (B,Nc,cap) int32 candidate lists, pos (B,Nc,G) int64 positions into them (-1 = the centre
    has no candidate) -> index (B,Nc,G) int64 and points (B,Nc,G,C); -1 / -1.0 where pos < 0."""
    _need_f32(pc, "pc")
    _need_i64(pos, "pos")
    if cand.dtype != torch.int32:
        raise TypeError("cand must be int32")
    if pc.stride(2) != 1:
        pc = pc.contiguous()
    cand, pos = cand.contiguous(), pos.contiguous()
    B, Nc, G = pos.shape
    C = pc.shape[2]
    with torch.cuda.device(pc.device):
        index = torch.empty((B, Nc, G), dtype=torch.int64, device=pc.device)
        points = torch.empty((B, Nc, G, C), dtype=torch.float32, device=pc.device)
        flag = _range_flag(pc.device)
        _check(_L.regnet_resample_groups_f32(pc.data_ptr(), pc.stride(0), pc.stride(1), C, cand.data_ptr(), cand.size(2),
                                             pos.data_ptr(), B, Nc, G, pc.shape[1], flag.data_ptr(), index.data_ptr(),
                                             points.data_ptr(), _stream(pc)), "resample_groups")
    return index, points


def _gather_max_arg(feature_rows, rows):
    R, G = rows.shape
    F = feature_rows.shape[1]
    with torch.cuda.device(feature_rows.device):
        out = torch.empty((R, F), dtype=torch.float32, device=feature_rows.device)
        arg = torch.empty((R, F), dtype=torch.int64, device=feature_rows.device)
        _check(_L.regnet_gather_max_arg_f32(feature_rows.data_ptr(), feature_rows.shape[0], F, rows.data_ptr(), R, G,
                                            out.data_ptr(), arg.data_ptr(), _stream(feature_rows)), "gather_max_arg")
    return out, arg


def _scatter_max_grad(dy, arg, grad, scene_rows, batch_stride, row_stride, ch_stride):
    dy = dy.contiguous()
    with torch.cuda.device(dy.device):
        _check(_L.regnet_scatter_max_grad_f32(dy.data_ptr(), arg.data_ptr(), arg.shape[0], arg.shape[1], scene_rows,
                                              batch_stride, row_stride, ch_stride, grad.data_ptr(), _stream(dy)),
               "scatter_max_grad")


class _GatherMaxFn(torch.autograd.Function):
    """gather_max with autograd (training): the backward scatters the R x F incoming values to the rows that gave the
    maxima (one launch of float atomics into a zeroed (n, F) gradient), instead of the reference's materialised (R, G, F)
    gather, its max and their zero-filled gradients."""

    @staticmethod
    def forward(ctx, feature_rows, rows):
        out, arg = _gather_max_arg(feature_rows, rows)
        ctx.save_for_backward(arg)
        ctx.src_shape = tuple(feature_rows.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        n, F = ctx.src_shape
        grad = torch.zeros((n, F), dtype=torch.float32, device=dy.device)
        _scatter_max_grad(dy, arg, grad, max(n, 1), 0, F, 1)
        return grad, None


# A trainer that already holds a gradient of the channel-first feature map (the segmentation head's, EARLY_HEAD_BACKWARD) sets
# it here for the duration of the region stage's backward: (tensor (B, F, N) contiguous float32, event after its last writer).
# The pools then ADD their gradient into it and return none of their own -- no zero-filled 210 MB gradients, no sum of the two
# pools', no transposed copy back to channel-first, no addition to the head's (0.7 ms of device time at 8 x 25 600).
_grad_sink = [None]


def set_feature_grad_sink(tensor, ready_event=None):
    """``tensor`` None: back to returning gradients through autograd.  See ``_grad_sink``."""
    _grad_sink[0] = None if tensor is None else (tensor, ready_event)


class _GatherMaxMapFn(torch.autograd.Function):
    """``gather_max`` from the feature map itself: ``feature_map`` (B, N, F) is any view of the network's (B, F, N) output,
    ``rows_copy`` its (B * N, F) contiguous copy made without autograd (the pools gather from it), rows (R, G) global row
    ids.  The backward writes the gradient in the map's own channel-first layout -- or into the trainer's sink."""

    @staticmethod
    def forward(ctx, feature_map, rows_copy, rows):
        out, arg = _gather_max_arg(rows_copy, rows)
        ctx.save_for_backward(arg)
        ctx.map_shape = tuple(feature_map.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        B, N, F = ctx.map_shape
        sink = _grad_sink[0]
        if sink is not None and tuple(sink[0].shape) == (B, F, N) and sink[0].is_contiguous() and sink[0].device == dy.device:
            if sink[1] is not None:
                torch.cuda.current_stream(dy.device).wait_event(sink[1])
            _scatter_max_grad(dy, arg, sink[0], N, F * N, 1, N)
            return None, None, None
        grad = torch.zeros((B, F, N), dtype=torch.float32, device=dy.device)
        _scatter_max_grad(dy, arg, grad, N, F * N, 1, N)
        return grad.transpose(1, 2), None, None


def gather_max_train(feature_rows, rows):
    """``gather_max`` for tensors that need gradients (GPU): (R_all,F) contiguous float32, rows (R,G) int64 -> (R,F)."""
    _need_f32(feature_rows, "feature_rows")
    _need_i64(rows, "rows")
    return _GatherMaxFn.apply(feature_rows.contiguous(), rows.contiguous())


def gather_max_map_train(feature_map, rows_copy, rows):
    """``gather_max`` for the (B, N, F) feature map of a training iteration, given its contiguous rows (no graph attached):
    -> (R, F); the gradient reaches ``feature_map`` directly (see _GatherMaxMapFn)."""
    _need_f32(rows_copy, "rows_copy")
    _need_i64(rows, "rows")
    return _GatherMaxMapFn.apply(feature_map, rows_copy, rows.contiguous())


def rowsum_neg(x, K):
    """-(sum over the last axis of K contiguous floats): x (..., K) contiguous float32 GPU -> x.shape[:-1]."""
    _need_f32(x, "x")
    x = x.contiguous()
    rows = x.numel() // K
    with torch.cuda.device(x.device):
        out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        _check(_L.regnet_rowsum_neg_f32(x.data_ptr(), rows, K, out.data_ptr(), _stream(x)), "rowsum_neg")
    return out


_range_flags = {}


def _range_flag(device):
    """Per-device int32 the kernels OR into when a drawn position / candidate is out of range (checked lazily: reading
    it is a synchronisation)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _range_flags:
        _range_flags[key] = torch.zeros((1,), dtype=torch.int32, device=device)
    return _range_flags[key]


def raise_if_out_of_range():
    """RuntimeError if any resample_groups launch since the last check saw a position beyond its candidate list or a
    candidate beyond the cloud (the torch.gather it replaced raised at once).  Synchronises; called where the caller
    synchronises anyway (end of pipeline.forward_scenes / ForwardPipeline.run)."""
    for flag in _range_flags.values():
        if int(flag.item()):
            flag.zero_()
            raise RuntimeError("resample_groups: a drawn position or candidate index was out of range")


def box_candidates(group_points, centre, rot, xlim, ylim, zlim):
    """group_points (n,G,C>=3), centre (n,3), rot (n,3,3), xlim/ylim (n) float32, zlim float ->
    cand (n,G) int32 ascending in-box positions, count (n) int32."""
    _need_f32(group_points, "group_points")
    n, G, _ = group_points.shape
    gp = group_points if group_points.stride(2) == 1 else group_points.contiguous()
    centre, rot = centre.contiguous().float(), rot.contiguous().float()
    xlim, ylim = xlim.contiguous().float(), ylim.contiguous().float()
    with torch.cuda.device(gp.device):
        cand = torch.empty((n, max(G, 1)), dtype=torch.int32, device=gp.device)
        count = torch.empty((n,), dtype=torch.int32, device=gp.device)
        _check(_L.regnet_box_crop_f32(gp.data_ptr(), gp.stride(0), gp.stride(1), centre.data_ptr(), rot.data_ptr(),
                                      xlim.data_ptr(), ylim.data_ptr(), float(zlim), n, G, cand.data_ptr(),
                                      count.data_ptr(), _stream(gp)), "box_crop")
    return cand, count


def gripper_frame(grasp):
    """grasp (n, >= 7) float32 GPU -> centre (n,3), rot (n,3,3) with rows [approach; axis_y; minor_normal]."""
    _need_f32(grasp, "grasp")
    g = grasp if grasp.stride(1) == 1 else grasp.contiguous()
    n = g.shape[0]
    with torch.cuda.device(g.device):
        centre = torch.empty((n, 3), dtype=torch.float32, device=g.device)
        rot = torch.empty((n, 3, 3), dtype=torch.float32, device=g.device)
        _check(_L.regnet_gripper_frame_f32(g.data_ptr(), g.stride(0) if n else 7, n, centre.data_ptr(), rot.data_ptr(),
                                           _stream(g)), "gripper_frame")
    return centre, rot


def stage2_decode(x_cls, x_reg, centres, templates, radius, sigmoid_tail):
    """x_cls (n,A), x_reg (n,A,C) contiguous float32, centres (n, >= 3) rows (any row stride), templates (A,4) float32 ->
    next_grasp (n,C): gripper_region_network.py:69-90 without labels, one launch (csrc/region.hip)."""
    _need_f32(x_cls, "x_cls")
    _need_f32(x_reg, "x_reg")
    n, A, C = x_reg.shape
    x_cls, x_reg, templates = x_cls.contiguous(), x_reg.contiguous(), templates.contiguous()
    c = centres if centres.stride(1) == 1 else centres.contiguous()
    with torch.cuda.device(x_reg.device):
        out = torch.empty((n, C), dtype=torch.float32, device=x_reg.device)
        _check(_L.regnet_stage2_decode_f32(x_cls.data_ptr(), x_reg.data_ptr(), A, C, c.data_ptr(), c.stride(0) if n else 3,
                                           templates.data_ptr(), float(radius), int(bool(sigmoid_tail)), n, out.data_ptr(),
                                           _stream(x_reg)), "stage2_decode")
    return out


def refine_decode(grasp, x_cls, x_reg, radius, score_thre):
    """grasp (m, >= C) rows, x_cls (m,2), x_reg (m,C) float32 -> final_grasp (m,C), flags (2,m) uint8 [class 1 | class 1 and
    score above the threshold]: gripper_region_network.py:201-215 without labels, one launch."""
    _need_f32(grasp, "grasp")
    m, C = x_reg.shape
    g = grasp if grasp.stride(1) == 1 else grasp.contiguous()
    x_cls, x_reg = x_cls.contiguous(), x_reg.contiguous()
    with torch.cuda.device(g.device):
        final = torch.empty((m, C), dtype=torch.float32, device=g.device)
        flags = torch.empty((2, m), dtype=torch.uint8, device=g.device)
        _check(_L.regnet_refine_decode_f32(g.data_ptr(), g.stride(0) if m else C, x_cls.data_ptr(), x_reg.data_ptr(), C,
                                           float(radius), float(score_thre), m, final.data_ptr(), flags.data_ptr(),
                                           _stream(g)), "refine_decode")
    return final, flags


def crop_pick(cand, pos, valid, group_index):
    """cand (n,G) int32, pos (n,R) int64, valid (n) bool, group_index (n,G) int64 -> index, index_inall (n,R) int64."""
    n, G = cand.shape
    R = pos.shape[1]
    cand, pos = cand.contiguous(), pos.contiguous()
    valid8 = valid.contiguous().view(torch.uint8)
    gi = group_index if group_index.stride(1) == 1 else group_index.contiguous()
    _need_i64(gi, "group_index")
    with torch.cuda.device(cand.device):
        index = torch.empty((n, R), dtype=torch.int64, device=cand.device)
        index_inall = torch.empty((n, R), dtype=torch.int64, device=cand.device)
        _check(_L.regnet_crop_pick(cand.data_ptr(), G, pos.data_ptr(), R, valid8.data_ptr(), gi.data_ptr(),
                                   gi.stride(0) if n else G, n, index.data_ptr(), index_inall.data_ptr(), _stream(cand)), "crop_pick")
    return index, index_inall


def gather_max_scene(feature_rows, index, row_ids, per_scene, scene_stride):
    """``gather_max`` with the scene offset formed in the address: feature_rows (B*N, F) contiguous (F % 4 == 0), index
    (R_all, G) int64 LOCAL point ids per centre, row_ids (R,) int64 or None (= every row of ``index``), scene of row id r
    = r // per_scene, its feature rows start at scene * scene_stride -> (R, F) max over G (negative ids are skipped)."""
    _need_f32(feature_rows, "feature_rows")
    _need_i64(index, "index")
    feature_rows, index = feature_rows.contiguous(), index.contiguous()
    G = index.shape[1]
    F = feature_rows.shape[1]
    if row_ids is not None:
        _need_i64(row_ids, "row_ids")
        row_ids = row_ids.contiguous()
    R = index.shape[0] if row_ids is None else row_ids.numel()
    with torch.cuda.device(feature_rows.device):
        out = torch.empty((R, F), dtype=torch.float32, device=feature_rows.device)
        _check(_L.regnet_gather_max_scene_f32(feature_rows.data_ptr(), feature_rows.shape[0], F, index.data_ptr(),
                                              None if row_ids is None else row_ids.data_ptr(), R, G, int(per_scene),
                                              int(scene_stride), out.data_ptr(), _stream(feature_rows)), "gather_max_scene")
    return out


def gather_max(feature_rows, rows):
    """feature_rows (R_all,F) contiguous, rows (R,G) int64 global row ids -> (R,F) max over G."""
    _need_f32(feature_rows, "feature_rows")
    _need_i64(rows, "rows")
    feature_rows, rows = feature_rows.contiguous(), rows.contiguous()
    R, G = rows.shape
    F = feature_rows.shape[1]
    with torch.cuda.device(feature_rows.device):
        out = torch.empty((R, F), dtype=torch.float32, device=feature_rows.device)
        _check(_L.regnet_gather_max_f32(feature_rows.data_ptr(), feature_rows.shape[0], F, rows.data_ptr(), R, G,
                                        out.data_ptr(), _stream(feature_rows)), "gather_max")
    return out
