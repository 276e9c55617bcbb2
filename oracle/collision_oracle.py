"""CPU ORACLE for the reference's grasp evaluation (test infrastructure, NOT product code).

Restates dataset_utils/eval_score/eval.py:4-24 -> eval_utils/evaluation_data_generator.py:
  * class EvalDataTest -- ``inv_transform_predicted_grasp`` (:109-170), the global->local matrices (:91-93) and
    ``finger_hand_view`` (:188-236), which test.py applies to every predicted grasp through utils.eval_notruth
    (utils.py:391-401);
  * class EvalDataValidate -- ``finger_hand_view`` (:420-483, table margin with the opposite sign and one more
    threshold), ``finger_hand_scene`` (:485-537), ``_antipodal_score`` (:392-418) and ``run_collision`` (:352-366), used
    on validation grasps through utils.eval_grasp_with_gt (utils.py:270-295).
Constants: eval_score/configs/config.py.  Citations are relative to /root/reference.

The VIEW cloud's estimated normals (open3d, :77-80 / :261-262), the kd-trees (:260, torch_scene_point_cloud.py:24) and
the table-corner tests (:172-186 / :374-386, results unused) do not influence anything these functions return and are
not restated; the SCENE cloud's normals are taken from the record (``scene_normal``, torch_scene_point_cloud.py:13-16)
or, for records without them, estimated by ``estimate_normals`` (torch_scene_point_cloud.py:17-19 ->
eval_utils/pointcloud.py:27-43).  That function calls open3d (``estimate_normals`` with ``KDTreeSearchParamHybrid``,
``normalize_normals``, ``orient_normals_towards_camera_location``), a dependency that is NOT in the reference tree and
not in this image (the reference pins no version; the method-style API it calls exists from open3d 0.8): PARITY UNPINNED
for ``estimate_normals`` -- it restates open3d's published algorithm (FLANN radius search with the squared radius
passed as a float and capped at max_nn nearest, covariance from the raw second moments in double, smallest-eigenvalue
eigenvector of a self-adjoint solver, (0,0,1) for fewer than three neighbours or a zero vector, flip towards the
camera) and is anchored only by known-answer surfaces (planes, spheres, isolated points) in tests/.

Canonical arithmetic of the point transform, shared with csrc/region.hip:grasp_collision_kernel: individually rounded
binary32 operations in source order, ``x = ((t00*px + t01*py) + t02*pz) + t03`` (the reference's 4xN torch.matmul goes
through a BLAS whose summation order / FMA use is unspecified; a point within one ulp of a box face may therefore fall
on the other side there).
"""
import warnings

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
CLOSE_REGION_MIN_POINTS = 16
NEIGHBOR_DEPTH = 0.005
TABLE_MARGIN = 0.005           # evaluation_data_generator.py:195 (+) and :428 (-)


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


def _local(points, t):
    p = np.ascontiguousarray(points, dtype=np.float32)
    px, py, pz = p[None, :, 0], p[None, :, 1], p[None, :, 2]

    def coord(r):
        return ((t[:, r, 0, None] * px + t[:, r, 1, None] * py) + t[:, r, 2, None] * pz) + t[:, r, 3, None]
    return coord(0), coord(1), coord(2)


def _regions(x, y, z, depth, width):
    f = np.float32
    x_lo = f(-BOTTOM_LENGTH)
    x_hi = np.asarray(depth, dtype=np.float32).reshape(-1, 1) if np.ndim(depth) else f(depth)
    hw, hs = f(width / 2 + FINGER_WIDTH), f(width / 2)
    th, bm = f(HALF_HAND_THICKNESS), f(-BACK_COLLISION_MARGIN)
    close = (x > x_lo) & (x < x_hi)
    zc = close & (z < th) & (z > -th)
    back = zc & (y < hw) & (y > -hw) & (x < bm)
    finger = zc & (((y < hw) & (y > hs)) | ((y > -hw) & (y < -hs)))
    region = zc & (y < hs) & (y > -hs)
    return close, back, finger, region


def collision_counts(points, T, depth, width, chunk=64):
    """points (N,3) float32, T (B,4,4) float32 -> int32 (B,4): points in the closing slab, behind the hand, inside a
    finger, between the fingers.  evaluation_data_generator.py:200-229 / :438-476 with the canonical arithmetic of the
    module docstring; Python scalars are compared as float32, as torch does for a float32 tensor against a Python
    number.  ``depth``: a float or one value per grasp (:428-430)."""
    T = np.ascontiguousarray(T, dtype=np.float32)
    out = np.zeros((T.shape[0], 4), dtype=np.int32)
    for s in range(0, T.shape[0], chunk):
        x, y, z = _local(points, T[s:s + chunk])
        d = np.asarray(depth, dtype=np.float32)[s:s + chunk] if np.ndim(depth) else depth
        for k, m in enumerate(_regions(x, y, z, d, width)):
            out[s:s + chunk, k] = m.sum(1)
    return out


def antipodal_scores(points, normals, T, depth, width):
    """float32 (B,): evaluation_data_generator.py:392-418 applied to the closing-region points of every grasp (:531-537):
    mean |n_y| over the points within d of the largest y times the same near the smallest y, d = min((y_max - y_min) / 3,
    NEIGHBOR_DEPTH), n_y the y component of the normal in the grasp frame.  NaN where the region is empty."""
    T = np.ascontiguousarray(T, dtype=np.float32)
    nrm = np.ascontiguousarray(normals, dtype=np.float32)
    out = np.full((T.shape[0],), np.nan, dtype=np.float32)
    for b in range(T.shape[0]):
        t = T[b:b + 1]
        x, y, z = _local(points, t)
        d = np.asarray(depth, dtype=np.float32)[b:b + 1] if np.ndim(depth) else depth
        region = _regions(x, y, z, d, width)[3][0]
        if not region.any():
            continue
        yy = y[0][region]
        ny = np.abs((t[0, 1, 0] * nrm[region, 0] + t[0, 1, 1] * nrm[region, 1]) + t[0, 1, 2] * nrm[region, 2])
        ly, ry = yy.max(), yy.min()
        dep = min(np.float32((ly - ry) / np.float32(3)), np.float32(NEIGHBOR_DEPTH))
        left, right = yy > ly - dep, yy < ry + dep
        with np.errstate(all="ignore"), warnings.catch_warnings():    # a one-point region has d = 0 and empty sides: NaN, like torch.mean of nothing
            warnings.simplefilter("ignore")
            out[b] = np.float32(ny[left].mean(dtype=np.float32)) * np.float32(ny[right].mean(dtype=np.float32))
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


def _passes(counts, with_region):
    ok = (counts[:, 0] >= NUM_POINTS_THRESHOLD) & ~(counts[:, 1] > BACK_COLLISION_THRESHOLD) & ~(counts[:, 2] > FINGER_COLLISION_THRESHOLD)
    return ok & (counts[:, 3] >= CLOSE_REGION_MIN_POINTS) if with_region else ok


def eval_validate(data, predicted_grasp, view_num, table_height, depth, width, gpu=-1):
    """eval.py:14-24 / EvalDataValidate.run_collision: -> (vgr, score, n_view, grasps without view collision, grasps
    without scene collision).  ``data``: dict with view_cloud (N1,3), scene_cloud (N2,3), scene_normal (N2,3)."""
    grasp = torch.as_tensor(predicted_grasp).float()
    if grasp.dim() == 3:                                                    # (B,4,4) frames (:273-275)
        frame, center = grasp[:, :3, :3].contiguous(), grasp[:, :3, 3].contiguous()
    else:
        frame, center = grasp_frames(grasp.view(-1, 8))
    T = global_to_local(frame, center).numpy()
    f = np.float32
    dep = np.asarray(depth, dtype=np.float32) if np.ndim(depth) else f(depth)
    view = collision_counts(np.asarray(data["view_cloud"], dtype=np.float32), T, depth, width)
    above = ~((center.numpy()[:, 2] + frame.numpy()[:, 2, 0] * dep) < f(table_height - TABLE_MARGIN))      # :427-432
    keep_view = np.nonzero(above & _passes(view, True))[0]
    Tv = T[keep_view]
    dv = np.asarray(depth, dtype=np.float32)[keep_view] if np.ndim(depth) else depth
    scene_pts = np.asarray(data["scene_cloud"], dtype=np.float32)
    scene = collision_counts(scene_pts, Tv, dv, width)
    ok = _passes(scene, True)
    score = np.zeros((len(keep_view),), dtype=np.float32)
    if ok.any():
        dvo = np.asarray(dv, dtype=np.float32)[ok] if np.ndim(dv) else dv
        score[ok] = antipodal_scores(scene_pts, np.asarray(data["scene_normal"], dtype=np.float32), Tv[ok], dvo, width)
    grasp_view = grasp[torch.from_numpy(keep_view)]
    return int(ok.sum()), float(torch.from_numpy(score).sum().item()), len(keep_view), grasp_view, grasp_view[torch.from_numpy(np.nonzero(ok)[0])]


NORMAL_RADIUS = 0.01           # eval_score/configs/config.py:16-17
NORMAL_MAX_NN = 30


def estimate_normals(points, camera_pos=(0.0, 0.0, 0.0), radius=NORMAL_RADIUS, max_nn=NORMAL_MAX_NN, chunk=512,
                     return_gap=False):
    """eval_utils/pointcloud.py:27-43 (see the header: open3d's algorithm restated, parity unpinned).
    points (N,3) -> (normals (N,3) float64 unit vectors facing the camera, neighbour counts (N,)).
    Brute force: every point against every point, squared distances in double in the order ((dx*dx)+(dy*dy))+(dz*dz),
    neighbours = those strictly below float(radius*radius), nearest ``max_nn`` ranked by (distance, index).
    ``return_gap`` adds (lambda_1 - lambda_0) / lambda_2 per point (0 where the normal is not an eigenvector): how well
    the smallest-eigenvalue direction is determined."""
    p = np.asarray(points, dtype=np.float32).astype(np.float64)     # the product path holds float32 coordinates
    N = p.shape[0]
    r2 = float(np.float32(float(radius) * float(radius)))
    cam = np.asarray(camera_pos, dtype=np.float64)
    normals = np.zeros((N, 3))
    counts = np.zeros((N,), dtype=np.int64)
    gap = np.zeros((N,))
    for beg in range(0, N, chunk):
        q = p[beg:beg + chunk]
        dx = p[None, :, 0] - q[:, None, 0]
        dy = p[None, :, 1] - q[:, None, 1]
        dz = p[None, :, 2] - q[:, None, 2]
        d2 = ((dx * dx) + (dy * dy)) + (dz * dz)
        for r in range(q.shape[0]):
            idx = np.nonzero(d2[r] < r2)[0]
            if idx.size > max_nn:
                idx = idx[np.lexsort((idx, d2[r, idx]))[:max_nn]]
            counts[beg + r] = idx.size
            n = np.array([0.0, 0.0, 1.0])
            if idx.size >= 3:
                nb = p[idx]
                mean = nb.mean(axis=0)
                cov = nb.T @ nb / idx.size - np.outer(mean, mean)      # raw second moments minus mean products
                w, v = np.linalg.eigh(cov)
                n = v[:, 0].copy()
                gap[beg + r] = (w[1] - w[0]) / w[2] if w[2] > 0 else 0.0
                if not n.any():
                    n = np.array([0.0, 0.0, 1.0])
            n = n / np.linalg.norm(n)
            if n @ (cam - q[r]) < 0:
                n = -n
            normals[beg + r] = n
    return (normals, counts, gap) if return_gap else (normals, counts)
