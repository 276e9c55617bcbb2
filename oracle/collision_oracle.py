"""CPU ORACLE for the view-cloud collision filter of test.py (test infrastructure, NOT product code).

Restates dataset_utils/eval_score/eval.py:4-12 -> eval_utils/evaluation_data_generator.py (class EvalDataTest):
``inv_transform_predicted_grasp`` (:109-170), the global->local matrices (:91-93) and ``finger_hand_view`` (:188-236),
which test.py applies to every predicted grasp through utils.eval_notruth (utils.py:391-401).  Constants:
eval_score/configs/config.py.  Citations are relative to /root/reference.

The estimated normals of that class (open3d, :77-80) and its table-corner test (:172-186, result unused at :197) do
not influence the returned grasps and are not restated.

Canonical arithmetic of the point transform, shared with csrc/region.hip:grasp_collision_kernel: individually rounded
binary32 operations in source order, ``x = ((t00*px + t01*py) + t02*pz) + t03`` (the reference's 4xN torch.matmul goes
through a BLAS whose summation order / FMA use is unspecified; a point within one ulp of a box face may therefore fall
on the other side there).
"""
import numpy as np
import torch

# eval_score/configs/config.py:9,25-28,36-40
NUM_POINTS_THRESHOLD = 16
BACK_COLLISION_THRESHOLD = 0.0
BACK_COLLISION_MARGIN = 0.0
FINGER_COLLISION_THRESHOLD = 0
FINGER_WIDTH = 0.01
HALF_HAND_THICKNESS = 0.005
BOTTOM_LENGTH = 0.06
TABLE_MARGIN = 0.005           # evaluation_data_generator.py:195


def grasp_frames(grasp):
    """(B,8) [centre(3), axis_y(3), angle, score] -> frame (B,3,3) with columns (approach, axis_y, minor normal),
    centre (B,3).  evaluation_data_generator.py:109-170, op for op (torch CPU)."""
    grasp = grasp.float().view(-1, 8)
    center = grasp[:, :3].contiguous()
    axis_y = grasp[:, 3:6]
    angle = grasp[:, 6]
    cos_t, sin_t = torch.cos(angle), torch.sin(angle)
    B = len(grasp)
    one, zero = torch.ones((B, 1)), torch.zeros((B, 1))
    R1 = torch.cat((cos_t.view(B, 1), zero, -sin_t.view(B, 1), zero, one, zero, sin_t.view(B, 1), zero, cos_t.view(B, 1)),
                   dim=1).view(B, 3, 3)
    norm_y = torch.norm(axis_y, dim=1)
    axis_y = torch.div(axis_y, norm_y.view(-1, 1))
    axis_y[torch.nonzero(torch.eq(norm_y, 0))] = torch.tensor([0, 1, 0], dtype=torch.float)
    axis_x = torch.cat((axis_y[:, 1].view(-1, 1), -axis_y[:, 0].view(-1, 1), zero), 1)
    norm_x = torch.norm(axis_x, dim=1)
    axis_x = torch.div(axis_x, norm_x.view(-1, 1))
    axis_x[torch.nonzero(torch.eq(norm_x, 0))] = torch.tensor([1, 0, 0], dtype=torch.float)
    axis_z = torch.cross(axis_x, axis_y, dim=1)
    norm_z = torch.norm(axis_z, dim=1)
    axis_z = torch.div(axis_z, norm_z.view(-1, 1))
    axis_z[torch.nonzero(torch.eq(norm_z, 0))] = torch.tensor([0, 0, 1], dtype=torch.float)
    matrix = torch.cat((axis_x.view(-1, 3, 1), axis_y.view(-1, 3, 1), axis_z.view(-1, 3, 1)), dim=2)
    matrix = torch.bmm(matrix, R1)
    approach = matrix[:, :, 0]
    norm_x = torch.norm(approach, dim=1)
    approach = torch.div(approach, norm_x.view(-1, 1))
    approach[torch.nonzero(torch.eq(norm_x, 0))] = torch.tensor([1, 0, 0], dtype=torch.float)
    minor_normal = torch.cross(approach, axis_y, dim=1)
    frame = torch.cat((approach.view(-1, 3, 1), axis_y.view(-1, 3, 1), minor_normal.view(-1, 3, 1)), dim=2).contiguous()
    return frame, center


def global_to_local(frame, center):
    """(B,4,4): rotation frame^T, translation -frame^T c.  evaluation_data_generator.py:91-93."""
    T = torch.eye(4).unsqueeze(0).expand(frame.shape[0], 4, 4).contiguous()
    T[:, 0:3, 0:3] = frame.transpose(1, 2)
    T[:, 0:3, 3:4] = -torch.bmm(frame.transpose(1, 2), center.unsqueeze(2))
    return T


def collision_counts(points, T, depth, width, chunk=64):
    """points (N,3) float32, T (B,4,4) float32 -> int32 (B,3): points in the closing slab, behind the hand, inside a
    finger.  evaluation_data_generator.py:200-229 with the canonical arithmetic of the module docstring; Python scalars
    are compared as float32, as torch does for a float32 tensor against a Python number."""
    p = np.ascontiguousarray(points, dtype=np.float32)
    T = np.ascontiguousarray(T, dtype=np.float32)
    f = np.float32
    x_lo, x_hi = f(-BOTTOM_LENGTH), f(depth)
    hw, hs = f(width / 2 + FINGER_WIDTH), f(width / 2)
    th, bm = f(HALF_HAND_THICKNESS), f(-BACK_COLLISION_MARGIN)
    px, py, pz = p[None, :, 0], p[None, :, 1], p[None, :, 2]
    out = np.zeros((T.shape[0], 3), dtype=np.int32)
    for s in range(0, T.shape[0], chunk):
        t = T[s:s + chunk]

        def coord(r):
            return ((t[:, r, 0, None] * px + t[:, r, 1, None] * py) + t[:, r, 2, None] * pz) + t[:, r, 3, None]
        x, y, z = coord(0), coord(1), coord(2)
        close = (x > x_lo) & (x < x_hi)
        zc = (z < th) & (z > -th)
        back = close & (y < hw) & (y > -hw) & (x < bm) & zc
        finger = close & zc & (((y < hw) & (y > hs)) | ((y > -hw) & (y < -hs)))
        out[s:s + chunk, 0] = close.sum(1)
        out[s:s + chunk, 1] = back.sum(1)
        out[s:s + chunk, 2] = finger.sum(1)
    return out


def accept(counts, frame, center, table_height, depth):
    """bool (B,): the grasps finger_hand_view keeps (:195-196, :203, :218, :229)."""
    f = np.float32
    c = np.asarray(center, dtype=np.float32)
    fr = np.asarray(frame, dtype=np.float32)
    above = ~((c[:, 2] + fr[:, 2, 0] * f(depth)) < f(table_height + TABLE_MARGIN))
    return (above & (counts[:, 0] >= NUM_POINTS_THRESHOLD) & ~(counts[:, 1] > BACK_COLLISION_THRESHOLD)
            & ~(counts[:, 2] > FINGER_COLLISION_THRESHOLD))


def eval_test(points, predicted_grasp, view_num, table_height, depth, width, gpu=-1):
    """eval.py:4-12: the predicted grasps (rows of (B,8)) that do not collide with the view cloud, in input order."""
    grasp = torch.as_tensor(predicted_grasp).float().view(-1, 8)
    if grasp.shape[0] == 0:
        return grasp
    frame, center = grasp_frames(grasp)
    T = global_to_local(frame, center)
    counts = collision_counts(np.asarray(torch.as_tensor(points).float()), T.numpy(), depth, width)
    keep = accept(counts, frame.numpy(), center.numpy(), table_height, depth)
    return grasp[torch.from_numpy(np.nonzero(keep)[0])]
