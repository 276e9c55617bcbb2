"""View-cloud collision filter of the predicted grasps (mirror of dataset_utils/eval_score/eval.py:4-12 ->
eval_utils/evaluation_data_generator.py, class EvalDataTest; applied by test.py:147 through utils.eval_notruth,
utils.py:391-401).

``eval_test(points, predicted_grasp, view_num, table_height, depth, width, gpu)`` keeps the reference's signature and
returns the rows of ``predicted_grasp[:, :8]`` whose gripper does not collide with the cloud, in input order.  The
reference loops over the grasps in Python (a 4xN matmul, five masks and a host sync per grasp); here the grasp frames
are computed batched with the reference's torch expressions and ONE kernel launch scans the cloud for all grasps
(csrc/region.hip:grasp_collision_kernel), followed by the reference's thresholds as tensor ops -- one sync in total.
GPU only (the reference's ``gpu=-1`` CPU mode is not offered: there is no CPU fallback in this package).

Not mirrored: ``eval_validate`` (antipodal scoring against the ground-truth scene cloud; needs open3d normal estimation).
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
TABLE_MARGIN = 0.005   # evaluation_data_generator.py:195


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


def collision_counts(points, T, depth, width):
    """points (N,3) float32 on the GPU (any strides), T (B,4,4) -> int32 (B,3): points in the closing slab / behind the
    hand / inside a finger for every grasp (evaluation_data_generator.py:200-229)."""
    if not points.is_cuda:
        raise RuntimeError("points must be a CUDA tensor (no CPU path)")
    if points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError("points must be float32 (N, 3)")
    T = T.to(points.device, torch.float32).contiguous()
    B, N = T.shape[0], points.shape[0]
    with torch.cuda.device(points.device):
        counts = torch.zeros((B, 3), dtype=torch.int32, device=points.device)
        _check(_L.regnet_grasp_collision_counts_f32(points.data_ptr(), points.stride(0), points.stride(1), N, T.data_ptr(),
                                                    B, -BOTTOM_LENGTH, float(depth), HALF_HAND_THICKNESS,
                                                    float(width) / 2 + FINGER_WIDTH, float(width) / 2,
                                                    -BACK_COLLISION_MARGIN, counts.data_ptr(),
                                                    torch.cuda.current_stream(points.device).cuda_stream),
               "grasp_collision_counts")
    return counts


def no_collision_mask(points, grasp, table_height, depth, width):
    """bool (B,): the grasps EvalDataTest.finger_hand_view keeps (:195-196, :203, :218, :229)."""
    frame, center = grasp_frames(grasp)
    counts = collision_counts(points, global_to_local(frame, center), depth, width)
    above = ~((center[:, 2] + frame[:, 2, 0] * depth) < (table_height + TABLE_MARGIN))
    return (above & (counts[:, 0] >= NUM_POINTS_THRESHOLD) & ~(counts[:, 1] > BACK_COLLISION_THRESHOLD)
            & ~(counts[:, 2] > FINGER_COLLISION_THRESHOLD))


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
