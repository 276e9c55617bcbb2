"""Grasp evaluation against point clouds (mirror of dataset_utils/eval_score/eval.py:4-24 ->
eval_utils/evaluation_data_generator.py, classes EvalDataTest / EvalDataValidate).

``eval_test(points, predicted_grasp, view_num, table_height, depth, width, gpu)`` -- test.py:147 through
utils.eval_notruth (utils.py:391-401) -- returns the rows of ``predicted_grasp[:, :8]`` whose gripper does not collide
with the view cloud, in input order.  ``eval_validate(formal_dict, predicted_grasp, view_num, table_height, depth, width,
gpu)`` -- utils.eval_grasp_with_gt (utils.py:270-295) -- additionally filters against the ground-truth scene cloud and
sums the antipodal scores: ``(vgr, score, n_view, grasps_view, grasps_scene)``.  Same signatures and return values.

The reference loops over the grasps in Python (a 4xN matmul, five or six masks and several host syncs per grasp, twice
for validation); here the grasp frames are computed batched with the reference's torch expressions and each pass is ONE
kernel launch over all grasps (csrc/region.hip: grasp_collision_kernel, grasp_antipodal_kernel) followed by the
reference's thresholds as tensor ops.  GPU only (the reference's ``gpu=-1`` CPU mode is not offered: there is no CPU
fallback in this package).  Scene normals are taken from the record (``scene_normal``); records without them get
``estimate_normals`` (open3d's hybrid-search PCA normals in the reference, torch_scene_point_cloud.py:17-19 ->
pointcloud.py:27-43; here a uniform-grid neighbour search + 3x3 eigen solve on the GPU, csrc/grid.hip).
"""
import torch

from . import _lib

_check = _lib.check
_L = _lib.lib

# dataset_utils/eval_score/configs/config.py:9,25-28,36-40
NUM_POINTS_THRESHOLD = 16
BACK_COLLISION_THRESHOLD = 0.0
BACK_COLLISION_MARGIN = 0.0
FINGER_COLLISION_THRESHOLD = 0
FINGER_WIDTH = 0.01
HALF_HAND_THICKNESS = 0.005
BOTTOM_LENGTH = 0.06
CLOSE_REGION_MIN_POINTS = 16
NEIGHBOR_DEPTH = 0.005
TABLE_MARGIN = 0.005   # evaluation_data_generator.py:195 (+, test flavour) and :428 (-, validation flavour)
NORMAL_RADIUS = 0.01   # configs/config.py:16-17
NORMAL_MAX_NN = 30


def _unit_or(v, fallback):
    """v / |v| with rows of zero norm replaced by ``fallback`` (the reference's div + nonzero(eq(norm, 0)) pattern)."""
    norm = torch.norm(v, dim=1)
    out = torch.div(v, norm.view(-1, 1))
    return torch.where((norm == 0).view(-1, 1), v.new_tensor(fallback).expand_as(out), out)


def grasp_frames(grasp):
    """(B,8) [centre(3), axis_y(3), angle, score] -> frame (B,3,3) with columns (approach, axis_y, minor normal) and
    centre (B,3): evaluation_data_generator.py:109-170, batched on the grasps' device."""
    grasp = grasp.float().view(-1, 8)
    B = grasp.shape[0]
    center = grasp[:, :3].contiguous()
    angle = grasp[:, 6]
    cos_t, sin_t = torch.cos(angle), torch.sin(angle)
    one, zero = grasp.new_ones((B, 1)), grasp.new_zeros((B, 1))
    R1 = torch.cat((cos_t.view(B, 1), zero, -sin_t.view(B, 1), zero, one, zero, sin_t.view(B, 1), zero, cos_t.view(B, 1)),
                   dim=1).view(B, 3, 3)
    axis_y = _unit_or(grasp[:, 3:6], [0.0, 1.0, 0.0])
    axis_x = _unit_or(torch.cat((axis_y[:, 1:2], -axis_y[:, 0:1], zero), 1), [1.0, 0.0, 0.0])
    axis_z = _unit_or(torch.cross(axis_x, axis_y, dim=1), [0.0, 0.0, 1.0])
    matrix = torch.bmm(torch.stack((axis_x, axis_y, axis_z), dim=2), R1)
    approach = _unit_or(matrix[:, :, 0], [1.0, 0.0, 0.0])
    minor_normal = torch.cross(approach, axis_y, dim=1)
    return torch.stack((approach, axis_y, minor_normal), dim=2).contiguous(), center


def global_to_local(frame, center):
    """(B,4,4) rotation frame^T, translation -frame^T c (evaluation_data_generator.py:91-93)."""
    T = torch.eye(4, device=frame.device).unsqueeze(0).repeat(frame.shape[0], 1, 1)
    T[:, 0:3, 0:3] = frame.transpose(1, 2)
    T[:, 0:3, 3:4] = -torch.bmm(frame.transpose(1, 2), center.unsqueeze(2))
    return T


def _points_ok(points, name="points"):
    if not points.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor (no CPU path)" % name)
    if points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError("%s must be float32 (N, 3)" % name)


def _depth_args(depth, B, device):
    """-> (scalar x_hi, per-grasp tensor or None): the reference accepts a float or one depth per grasp (:428-430)."""
    if isinstance(depth, torch.Tensor) and depth.dim() > 0:
        d = depth.to(device, torch.float32).contiguous().view(-1)
        if d.numel() != B:
            raise RuntimeError("one depth per grasp expected")
        return 0.0, d
    return float(depth), None


def _box_args(depth, width, B, device):
    x_hi, per = _depth_args(depth, B, device)
    return (-BOTTOM_LENGTH, x_hi, per.data_ptr() if per is not None else None, HALF_HAND_THICKNESS,
            float(width) / 2 + FINGER_WIDTH, float(width) / 2, -BACK_COLLISION_MARGIN), per


def collision_counts(points, T, depth, width):
    """points (N,3) float32 on the GPU (any strides), T (B,4,4) -> int32 (B,4): points in the closing slab / behind the
    hand / inside a finger / between the fingers, for every grasp (evaluation_data_generator.py:200-229, :438-476)."""
    _points_ok(points)
    T = T.to(points.device, torch.float32).contiguous()
    B, N = T.shape[0], points.shape[0]
    with torch.cuda.device(points.device):
        counts = torch.zeros((B, 4), dtype=torch.int32, device=points.device)
        box, keep = _box_args(depth, width, B, points.device)
        _check(_L.regnet_grasp_collision_counts_f32(points.data_ptr(), points.stride(0), points.stride(1), N, T.data_ptr(),
                                                    B, *box, counts.data_ptr(),
                                                    torch.cuda.current_stream(points.device).cuda_stream),
               "grasp_collision_counts")
    return counts


def antipodal_scores(points, normals, T, depth, width):
    """float32 (B,): product of the mean |n_y| near the two y extremes of every grasp's closing region
    (evaluation_data_generator.py:392-418 on the region of :521-534); meaningful where the region is not empty."""
    _points_ok(points)
    _points_ok(normals, "normals")
    T = T.to(points.device, torch.float32).contiguous()
    B, N = T.shape[0], points.shape[0]
    with torch.cuda.device(points.device):
        stats = torch.zeros((B, 4), dtype=torch.float32, device=points.device)
        sides = torch.zeros((B, 2), dtype=torch.int32, device=points.device)
        box, keep = _box_args(depth, width, B, points.device)
        _check(_L.regnet_grasp_antipodal_stats_f32(points.data_ptr(), points.stride(0), points.stride(1), normals.data_ptr(),
                                                   normals.stride(0), normals.stride(1), N, T.data_ptr(), B, *box,
                                                   NEIGHBOR_DEPTH, stats.data_ptr(), sides.data_ptr(),
                                                   torch.cuda.current_stream(points.device).cuda_stream),
               "grasp_antipodal_stats")
    return (stats[:, 2] / sides[:, 0]) * (stats[:, 3] / sides[:, 1])


def _passes(counts, with_region):
    ok = (counts[:, 0] >= NUM_POINTS_THRESHOLD) & ~(counts[:, 1] > BACK_COLLISION_THRESHOLD) \
        & ~(counts[:, 2] > FINGER_COLLISION_THRESHOLD)
    return ok & (counts[:, 3] >= CLOSE_REGION_MIN_POINTS) if with_region else ok


def no_collision_mask(points, grasp, table_height, depth, width):
    """bool (B,): the grasps EvalDataTest.finger_hand_view keeps (:195-196, :203, :218, :229)."""
    frame, center = grasp_frames(grasp)
    counts = collision_counts(points, global_to_local(frame, center), depth, width)
    above = ~((center[:, 2] + frame[:, 2, 0] * depth) < (table_height + TABLE_MARGIN))
    return above & _passes(counts, False)


def eval_test(points, predicted_grasp, view_num, table_height, depth, width, gpu=0):
    """points (N,3), predicted_grasp (B,8) -> the collision-free grasps (B',8), input order (eval.py:4-12).  ``view_num``
    only selects the camera towards which the reference orients its (unused) normals."""
    if gpu == -1:
        raise RuntimeError("eval_test: this package has no CPU mode (gpu=-1)")
    dev = torch.device("cuda", int(gpu))
    grasp = torch.as_tensor(predicted_grasp).float().to(dev).view(-1, 8)
    if grasp.shape[0] == 0:
        return grasp
    pts = torch.as_tensor(points).float().to(dev)
    return grasp[no_collision_mask(pts, grasp, table_height, depth, width)]


def estimate_normals(points, camera_pos=(0.0, 0.0, 0.0), radius=NORMAL_RADIUS, max_nn=NORMAL_MAX_NN, return_count=False):
    """PointCloud.estimate_normals (pointcloud.py:27-43): unit normals (N,3) float32 of the cloud ``points`` (N,3) on the
    GPU -- for every point the eigenvector of the smallest eigenvalue of the covariance of its ``max_nn`` nearest
    neighbours within ``radius`` (itself included; (0,0,1) when fewer than 3), facing ``camera_pos``.
    Caveat: the neighbourhood test runs on float32 coordinates (the cloud is cast before the radius test), whereas the
    reference's open3d path keeps float64 points; a neighbour whose distance is within float32 rounding of the radius
    can fall on the other side."""
    if not points.is_cuda:
        raise RuntimeError("estimate_normals: points must be on the GPU (no CPU fallback)")
    pts = points.float().contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("estimate_normals: points must be (N,3)")
    N = pts.shape[0]
    cam = [float(c) for c in camera_pos]
    with torch.cuda.device(pts.device):
        normals = torch.empty((N, 3), dtype=torch.float32, device=pts.device)
        count = torch.empty((N,), dtype=torch.int32, device=pts.device)
        ws = torch.empty((max(int(_L.regnet_normals_workspace_bytes(N)), 16),), dtype=torch.uint8, device=pts.device)
        _check(_L.regnet_estimate_normals_f32(pts.data_ptr(), N, float(radius), int(max_nn), cam[0], cam[1], cam[2],
                                              normals.data_ptr(), count.data_ptr(), ws.data_ptr(),
                                              torch.cuda.current_stream(pts.device).cuda_stream), "estimate_normals")
    return (normals, count) if return_count else normals


def eval_validate(formal_dict, predicted_grasp, view_num, table_height, depth, width, gpu=0):
    """eval.py:14-24 / EvalDataValidate.run_collision (:352-366).  ``formal_dict``: a validation record with
    ``view_cloud`` (N1,3), ``scene_cloud`` (N2,3) and optionally ``scene_normal`` (N2,3) (estimated from the scene
    cloud when absent, torch_scene_point_cloud.py:13-19); ``predicted_grasp`` (B,8) rows or (B,4,4) frames.  -> (vgr, antipodal score sum, grasps without view collision (count), those grasps, the ones that also clear
    the scene cloud)."""
    if gpu == -1:
        raise RuntimeError("eval_validate: this package has no CPU mode (gpu=-1)")
    dev = torch.device("cuda", int(gpu))
    grasp = torch.as_tensor(predicted_grasp).float().to(dev)
    if grasp.dim() == 3:                                                    # (B,4,4) frames (:273-275)
        frame, center = grasp[:, :3, :3].contiguous(), grasp[:, :3, 3].contiguous()
    else:
        grasp = grasp.view(-1, 8)
        frame, center = grasp_frames(grasp)
    if grasp.shape[0] == 0:
        return 0, 0.0, 0, grasp, grasp
    T = global_to_local(frame, center)
    dep = depth.to(dev, torch.float32).view(-1) if isinstance(depth, torch.Tensor) and depth.dim() > 0 else depth
    view = collision_counts(torch.as_tensor(formal_dict["view_cloud"]).float().to(dev), T, dep, width)
    above = ~((center[:, 2] + frame[:, 2, 0] * dep) < (table_height - TABLE_MARGIN))              # :427-432
    keep_view = torch.nonzero(above & _passes(view, True)).view(-1)
    grasp_view = grasp[keep_view]
    if keep_view.numel() == 0:
        return 0, 0.0, 0, grasp_view, grasp_view
    Tv = T[keep_view]
    dv = dep[keep_view] if isinstance(dep, torch.Tensor) else dep
    scene = torch.as_tensor(formal_dict["scene_cloud"]).float().to(dev)
    ok = _passes(collision_counts(scene, Tv, dv, width), True)
    if "scene_normal" in formal_dict:
        normal = torch.as_tensor(formal_dict["scene_normal"]).float().to(dev)
    else:
        normal = estimate_normals(scene)
    score = antipodal_scores(scene, normal, Tv, dv, width)
    score = torch.where(ok, score, torch.zeros_like(score))
    return int(ok.sum()), float(score.sum().item()), int(keep_view.numel()), grasp_view, grasp_view[torch.nonzero(ok).view(-1)]
