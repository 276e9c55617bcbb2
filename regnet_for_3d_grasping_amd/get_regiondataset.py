"""Region grouping between ScoreNet and the grasp-region network (mirror of
dataset_utils/get_regiondataset.py:13-42, :279-295, :311-434).

``get_grasp_allobj`` keeps the reference's signature and return tuple.  What changes is where
the work happens: centre selection and the radius scan run on the GPU (FPS kernel, radius-group
kernel producing ascending candidate lists + counts); the host only draws the
``np.random.choice`` positions, in exactly the reference's call order (all scenes x centres of
the small-radius pass first, then the large-radius pass -- get_regiondataset.py:36-37), so a
seeded numpy RNG yields the same groups.  One device->host sync per pass instead of one per
centre.

Training labels (``_get_center_grasp`` / ``_transform_grasp``, get_regiondataset.py:45-199): every
centre is matched to the nearest ground-truth grasp and the 4x4 frame is re-expressed as
(centre, closing axis, angle, scores).  ``data_paths`` holds one entry per scene -- a path to the
dataset's pickled dict or the dict itself (keys ``frame`` + ``antipodal_score``, or the ``select_*`` set).
"""
import numpy as np
import torch

from . import host_io, np_random, region_ops
from .pn2_utils import function as _F


def get_grasp_allobj(pc, predict_score, params, data_paths, use_theta=True, defer_large_groups=False):
    """pc (B,N,6), predict_score (B,N), params = [center_num, score_thre, group_num, r_time_group,
    group_num_more, r_time_group_more, width, height, depth] ->
    (center_pc (B,Nc,6), center_pc_index (B,Nc), pc_group_index (B,Nc,G), pc_group (B,Nc,G,6),
     pc_group_more_index (B,Nc,Gm), pc_group_more (B,Nc,Gm,6), grasp_labels=None)."""
    (center_num, score_thre, group_num, r_time_group, group_num_more, r_time_group_more,
     width, height, depth) = params
    center_pc, center_pc_index = _select_score_center(pc, predict_score, center_num, score_thre)
    labels_pending = None
    if pc.is_cuda and (DEVICE_DRAWS or (pc.shape[0] * pc.shape[1] > BATCHED_SEARCH_MAX_POINTS and len(data_paths) == 0)):
        # (large inference batches are throughput-bound: two candidate searches back to back at the start of the stage take CUs
        # from the next batch's first chain kernel -- same step time, but that launch read 4 % longer; they keep one search per
        # group.  With labels -- a training iteration -- the stage is the critical path of the host: one read, whatever the size)
        pc_group_index, pc_group = _get_group_pc(pc, center_pc, center_pc_index, group_num, width, height, depth,
                                                 r_time_group)
        pc_group_more_index, pc_group_more = _get_group_pc(pc, center_pc, center_pc_index, group_num_more, width,
                                                           height, depth, r_time_group_more)
    else:
        # host draws: BOTH candidate searches first, ONE device->host read for both count tables, then the draws in the
        # reference's order (all small groups, then all large ones) -- the first resampling runs on the device while the
        # host draws for the second (one synchronisation and ~0.3 ms of waiting less per batch than group by group)
        cand_s, count_s = region_ops.radius_candidates(pc, center_pc, group_radius(width, height, depth, r_time_group))
        cand_m, count_m = region_ops.radius_candidates(pc, center_pc, group_radius(width, height, depth, r_time_group_more))
        if len(data_paths) > 0 and pc.is_cuda:
            # the labels draw nothing from numpy's stream: their host preparation and launch go here, while the device is
            # still searching (the read of the counts below would just wait for it)
            labels_pending = _get_center_grasp(center_pc_index, center_pc, data_paths, depth, use_theta, defer_read=True)
        np_random.flush()
        counts = torch.stack((count_s, count_m)).cpu().numpy()
        def drawn(count, size):
            # on a GPU the positions (1 MB for the small groups of 8 x 64 centres, 8 MB for the large ones) are drawn straight into
            # page-locked memory and leave by DMA behind the draws; a pageable source costs the launching thread a staging copy
            # (1 ms for the large groups -- inside the region stage, the host-paced part of a training iteration)
            if pc.is_cuda:
                return np_random.choice_rows_pinned(count, size, 0)[0].to(pc.device, non_blocking=True)
            return host_io.upload(np_random.choice_rows(count, size, 0)[0], pc.device)

        pos = drawn(counts[0], group_num)
        pc_group_index, pc_group = region_ops.resample_groups(pc, cand_s, pos)

        def large_groups():
            return region_ops.resample_groups(pc, cand_m, drawn(counts[1], group_num_more))

        if defer_large_groups and DEFER_LARGE_GROUPS and pc.is_cuda:
            # (train_step) the large groups are first needed by the refine stage; their draws -- the next ones on numpy's stream
            # whoever makes them -- are left to the caller, who makes them while the device runs the region head:
            # ``pc_group_more_index`` is the callable, ``pc_group_more`` None (GripperRegionNetwork.forward resolves them)
            pc_group_more_index, pc_group_more = large_groups, None
        else:
            pc_group_more_index, pc_group_more = large_groups()
    grasp_labels = None
    if labels_pending is not None:
        grasp_labels = labels_pending() if callable(labels_pending) else labels_pending
    elif len(data_paths) > 0:
        grasp_labels = _get_center_grasp(center_pc_index, center_pc, data_paths, depth, use_theta)
    np_random.flush_unless_deferred()   # numpy's generator gets the state the device draws left (pipeline: once per run)
    return center_pc, center_pc_index, pc_group_index, pc_group, pc_group_more_index, pc_group_more, grasp_labels


DEFER_LARGE_GROUPS = True      # get_grasp_allobj(defer_large_groups=True) may leave the large groups' draws + resampling to its caller
LABEL_KERNEL = True            # GPU, batched: matching and _transform_grasp as one kernel (csrc/losses.hip: label_match_kernel)
BATCHED_LABELS = True          # GPU: match all scenes' centres to their grasp labels in one batched pass (_get_center_grasp)
NO_GRASP_SQ_DISTANCE = 0.005   # a centre further than this (SQUARED distance) from every grasp has no label (:120)


def _load_grasp_record(entry):
    """One scene's ground-truth grasps -> (frames (G,4,4), score, antipodal, centre-score) float32
    tensors (get_regiondataset.py:66-86).  With only ``frame`` / ``antipodal_score`` present all three
    scores are the antipodal score."""
    data = entry if isinstance(entry, dict) else np.load(entry, allow_pickle=True)

    def as_tensor(v):
        return torch.tensor(np.asarray(v), dtype=torch.float32) if not isinstance(v, torch.Tensor) else v.float()

    if "frame" in data.keys():
        frames = as_tensor(data["frame"])
        score = as_tensor(data["antipodal_score"])
        return frames, score, score, score
    return (as_tensor(data["select_frame"]), as_tensor(data["select_antipodal_score"]),
            as_tensor(data["select_antipodal_score"]), as_tensor(data["select_center_score"]))


def _compute_distance(points1, points2):
    """Squared distances (len(points1), len(points2)) by the expansion -2ab + |b|^2 + |a|^2, returned
    as float64 like the reference (get_regiondataset.py:271-277)."""
    a = points1[:, :3]
    d = -2 * a.mm(points2.transpose(1, 0))
    d = d + torch.sum(points2 * points2, 1).view(1, -1)
    d = d + torch.sum(a * a, 1).view(-1, 1)
    return d.double()


def _get_center_grasp(center_pc_index, center_pc, data_paths, depth, use_theta=True, defer_read=False):
    """Match every centre to its nearest ground-truth grasp (get_regiondataset.py:45-134).
    Returns (B, Nc, 10) = centre(3) | closing axis(3) | angle | score | antipodal | centre-score with
    -1 rows for centres without a grasp; (B, Nc, 13) raw-frame layout when ``use_theta`` is False."""
    B, Nc = center_pc_index.shape
    dev = center_pc.device
    label = torch.full((B, Nc, 3, 4), -1.0, device=dev)
    score_l = torch.full((B, Nc), -1.0, device=dev)
    anti_l = torch.full((B, Nc), -1.0, device=dev)
    cen_l = torch.full((B, Nc), -1.0, device=dev)
    # every scene's label arrays go up in ONE host->device copy (four pageable copies per scene, each a
    # synchronisation, before)
    loaded = [_load_grasp_record(entry) for entry in data_paths]
    if BATCHED_LABELS and dev.type == "cuda" and B > 1 and min(int(rec[0].shape[0]) for rec in loaded) > 0:
        # one batched match for all scenes (the per-scene loop below is ~30 small launches per scene on the host-paced
        # path of the training iteration, DESIGN.md par. 12.5): records padded to the longest, padded grasps at distance
        # +inf.  Same expression per element as _compute_distance, same arg-min.
        Gs = [int(rec[0].shape[0]) for rec in loaded]
        Gmax = max(Gs)
        host = np.zeros((B, Gmax, 19), dtype=np.float32)          # 16 frame entries | score | antipodal | centre score
        for i, (frames, score, anti, cen) in enumerate(loaded):
            host[i, :Gs[i], :16] = frames.reshape(Gs[i], 16).numpy()
            host[i, :Gs[i], 16], host[i, :Gs[i], 17], host[i, :Gs[i], 18] = score.numpy(), anti.numpy(), cen.numpy()
        if LABEL_KERNEL and use_theta:
            # matching + the (centre, axis, angle, scores) form of the match in ONE launch (csrc/losses.hip: label_match_kernel):
            # one host->device copy (records + counts), one launch, one read (does any row carry an antipodal score?)
            from . import _lib
            blob = np.empty((B * Gmax * 19 + B,), dtype=np.float32)
            blob[:B * Gmax * 19] = host.reshape(-1)
            blob[B * Gmax * 19:].view(np.int32)[:] = np.asarray(Gs, dtype=np.int32)
            packed = host_io.upload(blob, dev)
            out = torch.empty((B, Nc, 10), dtype=torch.float32, device=dev)
            wide_row = torch.empty((B * Nc,), dtype=torch.int32, device=dev)
            xyz = center_pc if center_pc.dtype == torch.float32 and center_pc.stride(2) == 1 else center_pc.float().contiguous()
            with torch.cuda.device(dev):
                _lib.check(_lib.lib.regnet_label_match_f32(
                    packed.data_ptr(), packed.data_ptr() + 4 * B * Gmax * 19, Gmax, xyz.data_ptr(), xyz.stride(0), xyz.stride(1),
                    B, Nc, float(np.float32(depth)), float(NO_GRASP_SQ_DISTANCE), out.data_ptr(), wide_row.data_ptr(),
                    torch.cuda.current_stream(dev).cuda_stream), "label_match")
            def finish():
                return out if bool(wide_row.cpu().numpy().any()) else out[:, :, :8].contiguous()

            return finish if defer_read else finish()     # (defer_read: the caller makes the one read when it suits it)
        packed, valid = host_io.upload_many((host, np.arange(Gmax)[None, :] < np.asarray(Gs)[:, None]), dev)
        frames = packed[:, :, :16].view(B, Gmax, 4, 4)
        approach = frames[:, :, :3, 0]
        contact = ((frames[:, :, :3, 3] + approach * depth).float() - approach * depth).float()
        a = center_pc[:, :, :3]
        d = -2 * torch.bmm(a, contact.transpose(1, 2))
        d = d + torch.sum(contact * contact, 2).view(B, 1, Gmax)
        d = d + torch.sum(a * a, 2).view(B, Nc, 1)
        d = d.double().masked_fill(~valid.view(B, 1, Gmax), float("inf"))
        dist, nearest = torch.min(d, dim=2)                           # (B, Nc)
        has = dist <= NO_GRASP_SQ_DISTANCE
        picked = torch.gather(packed, 1, nearest.unsqueeze(-1).expand(B, Nc, 19))
        picked = picked.masked_fill(~has.unsqueeze(-1), -1.0)
        label = picked[:, :, :16].view(B, Nc, 4, 4)[:, :, :3, :4]
        score_l, anti_l, cen_l = picked[:, :, 16], picked[:, :, 17], picked[:, :, 18]
        loaded = []                                                   # (skips the per-scene loop)
    flat = torch.cat([t.reshape(-1) for rec in loaded for t in rec]).to(dev) if loaded else None
    at = 0
    for i, rec in enumerate(loaded):
        parts = []
        for t in rec:
            parts.append(flat[at:at + t.numel()].view(t.shape))
            at += t.numel()
        frames, score, anti, cen = parts
        approach = frames[:, :3, 0]
        # the reference shifts the grasp centre along the approach by `depth` and back again (:91-92)
        contact = ((frames[:, :3, 3] + approach * depth).float() - approach * depth).float()
        dist, nearest = torch.min(_compute_distance(center_pc[i], contact), dim=1)
        has = dist <= NO_GRASP_SQ_DISTANCE
        minus = torch.full((), -1.0, device=dev)
        label[i] = torch.where(has.view(-1, 1, 1), frames[nearest, :3, :4], minus)
        score_l[i] = torch.where(has, score[nearest], minus)
        anti_l[i] = torch.where(has, anti[nearest], minus)
        cen_l[i] = torch.where(has, cen[nearest], minus)
    if use_theta:
        return _transform_grasp(label, score_l, anti_l, cen_l)
    flat = label.view(-1, 3, 4).clone()
    flip = flat[:, 0, 1] < 0
    flat[flip, :, 1:2] = -flat[flip, :, 1:2]
    out = torch.full((B, Nc, 13), -1.0, device=dev)
    out[:, :, :12] = flat.transpose(2, 1).contiguous().view(B, Nc, 12)
    out[:, :, 12] = score_l
    return out


def _wrap_angle(theta):
    """The reference's four in-place wrap steps (get_regiondataset.py:163-166), in its order."""
    two_pi = 2 * np.pi
    theta = torch.where(theta >= two_pi, theta - two_pi, theta)
    theta = torch.where(theta <= -two_pi, theta + two_pi, theta)
    theta = torch.where(theta > np.pi, theta - two_pi, theta)
    theta = torch.where(theta <= -np.pi, theta + two_pi, theta)
    return theta


def _transform_grasp(grasp_ori, grasp_score_ori, antipodal_score_ori, center_score_ori):
    """(B,Nc,3,4) frames [x|y|z|c] -> (centre, y axis with y.x >= 0, angle = atan2(x_z, z_z) [mirrored
    to pi - angle when y was flipped, wrapped to (-pi, pi]], scores)  (get_regiondataset.py:136-199).
    8 channels when no antipodal score is present at all, else 10."""
    B, Nc = grasp_score_ori.shape
    wide = bool((antipodal_score_ori != -1).any())
    axis_x = grasp_ori[:, :, :3, 0].reshape(B * Nc, 3)
    axis_y = grasp_ori[:, :, :3, 1].reshape(B * Nc, 3)
    axis_z = grasp_ori[:, :, :3, 2].reshape(B * Nc, 3)
    missing = (axis_x == -1).all(dim=1)
    angle = torch.atan2(axis_x[:, 2], axis_z[:, 2])
    flip = axis_y[:, 0] < 0
    angle = torch.where(flip, np.pi - angle, angle)
    axis_y = torch.where(flip.view(-1, 1), -axis_y, axis_y)
    angle = _wrap_angle(angle)
    angle = torch.where(missing, torch.full_like(angle, -1.0), angle)
    out = torch.full((B, Nc, 10 if wide else 8), -1.0, device=grasp_ori.device)
    out[:, :, :3] = grasp_ori[:, :, :3, 3]
    out[:, :, 3:6] = axis_y.view(B, Nc, 3)
    out[:, :, 6] = angle.view(B, Nc)
    out[:, :, 7] = grasp_score_ori
    if wide:
        out[:, :, 8] = antipodal_score_ori
        out[:, :, 9] = center_score_ori
    return out


_FPS_PAD_MIN = 1024   # scenes with more positives than this share one padded sampling launch


def _select_score_center(pc, pre_score, center_num, score_thre):
    """Pick ``center_num`` grasp centres per scene among points scoring > ``score_thre``
    (get_regiondataset.py:354-434): FPS over the positive subset when there are more than
    ``center_num`` positives (its first positive point is always centre 0); all positives padded
    with random repeats when 0 < P <= center_num; random points when P == 0.  The B == 1 and
    B > 1 branches of the reference implement the same rule and draw the same numpy variates.

    One host sync for the whole batch (the positive counts).  The ascending positive ids and their
    coordinates come from one compaction kernel, and the scenes that need sampling share ONE
    furthest-point-sampling launch over a common prefix length: the compacted coordinates are padded
    with copies of the scene's first positive, which is the sampler's start point -- the copies stay at
    distance 0 from the selected set and are never picked, and above 512 points the reference's tie
    order does not depend on the length, so each scene gets exactly its own per-scene result."""
    B, N, C = pc.shape
    score = pre_score.to(pc.device)
    if score.dtype != torch.float32:
        score = score.float()
    order, sub_xyz, count = region_ops.select_positive(pc, score.view(B, N), score_thre)
    counts = count.cpu().tolist()
    index = torch.empty((B, center_num), dtype=torch.int64, device=pc.device)
    # scenes sampled together: enough positives for the padding argument above to hold
    batched = [b for b in range(B) if counts[b] > max(center_num, _FPS_PAD_MIN)]
    if len(batched) > 1:
        pmax = max(counts[b] for b in batched)
        rows = torch.tensor(batched, device=pc.device)
        xyz_b = sub_xyz[:, :, :pmax] if len(batched) == B else sub_xyz[rows][:, :, :pmax]
        local = _F.farthest_point_sample(xyz_b, center_num)                  # (len(batched), center_num)
        picked = torch.gather(order if len(batched) == B else order[rows], 1, local)
        if len(batched) == B:
            index.copy_(picked)
        else:
            index[rows] = picked
    else:
        batched = []
    for b in range(B):
        if b in batched:
            continue
        P = int(counts[b])
        map_index = order[b, :P]
        if P > center_num:
            index[b] = map_index[_F.farthest_point_sample(sub_xyz[b:b + 1, :, :P], center_num).view(-1)]
        elif P > 0:
            np_random.flush()    # a host-side draw: numpy's generator must hold the current state
            extra = np.random.choice(P, center_num - P, replace=True)
            local = torch.cat([torch.arange(P), torch.from_numpy(np.asarray(extra, dtype=np.int64))])
            index[b] = map_index[local.to(pc.device)]
        else:
            np_random.flush()
            picks = np.random.choice(N, center_num, replace=False)
            index[b] = host_io.upload(np.asarray(picks, dtype=np.int64), pc.device)
    if pc.is_cuda and pc.dtype == torch.float32:
        # rows of the (B, N, C) cloud = gather_points with the roles of the axes swapped (one native launch)
        from . import pn2_ext
        center_pc = pn2_ext.gather_points(pc.transpose(1, 2), index, channels_last=True)
    else:
        center_pc = torch.gather(pc, 1, index.unsqueeze(-1).expand(B, center_num, C))
    return center_pc, index


def group_radius(width, height, depth, r_time):
    """Radius as the reference's comparison sees it: the Python double ``max(w,h,d)*r_time``
    compared against float32 distances, i.e. rounded to float32 (get_regiondataset.py:291-294)."""
    return float(np.float32(max(width, height, depth) * r_time))


# True: the resampling draws are made on the device from a device-resident copy of numpy's generator (no host round trip;
# csrc/np_random_dev.hip).  Measured (round 3, 8 x 25 600 points, region stage alone): the serial walk over numpy's stream
# costs 0.83 + 2.37 ms per batch on one workgroup against 1.2 ms of native host draws, so the host path stays the default;
# the device path is bit-identical (tests/test_gpu_np_random.py) and is what a host-bound deployment would switch on.
DEVICE_DRAWS = False
BATCHED_SEARCH_MAX_POINTS = 4 * 25600   # up to here get_grasp_allobj runs both candidate searches before its one read of the counts


def _draw_positions(counts, group_num, max_count):
    """Resampling of every (scene, centre) candidate list to exactly ``group_num`` entries, consuming numpy's
    global RNG in the reference's order (get_regiondataset.py:331-337: scene-major, then centre; without
    replacement when ``n >= group_num`` else with).  counts: (B,Nc) int32 tensor.  Returns positions
    (B,Nc,group_num) int64 into the ascending candidate lists; rows with no candidate are -1.
    On the GPU the draws are made by kernels from a device-resident copy of numpy's generator state
    (np_random.choice_rows_device: no synchronisation); ``DEVICE_DRAWS = False`` or CPU tensors (the oracle-backed
    mirror) take the native host code (np_random.choice_rows) after one device->host copy of the counts."""
    if DEVICE_DRAWS and counts.is_cuda:
        return np_random.choice_rows_device(counts.int(), group_num, 0, max_count)[0]
    np_random.flush()
    if counts.is_cuda:   # drawn into page-locked memory, copied by DMA behind the draws (1 + 4 MB per batch of 8)
        return np_random.choice_rows_pinned(counts.cpu().numpy(), group_num, 0)[0].to(counts.device, non_blocking=True)
    return host_io.upload(np_random.choice_rows(counts.cpu().numpy(), group_num, 0)[0], counts.device)


def _get_group_pc(pc, center_pc, center_pc_index, group_num, width, height, depth, r_time):
    """Radius grouping around every centre (get_regiondataset.py:311-352): candidates are the
    points with ``sqrt(dx^2+dy^2+dz^2) <= R`` (inclusive) in ascending index order, resampled to
    exactly ``group_num`` (without replacement when enough, else with).
    Returns pc_group_index (B,Nc,G) int64 and pc_group (B,Nc,G,C); empty groups stay -1."""
    B, N, C = pc.shape
    Nc = center_pc.shape[1]
    radius = group_radius(width, height, depth, r_time)
    cand, counts = region_ops.radius_candidates(pc, center_pc, radius)  # (B,Nc,cap) int32, (B,Nc) int32
    pos = _draw_positions(counts, group_num, N)                         # on the GPU: no synchronisation
    # candidate slots of an empty group were never written: the kernel does not read through them
    pc_group_index, pc_group = region_ops.resample_groups(pc, cand, pos)
    return pc_group_index, pc_group
