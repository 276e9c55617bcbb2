"""CPU ORACLE binding (test infrastructure, NOT product code).

Exposes the reference's ``pn2_ext`` / ``dgcnn_ext`` pybind surface
(multi_model/utils/pn2_utils/csrc/main.cpp:6-14, functions/csrc/main.cpp:3-6) on CPU torch
tensors, backed by ``oracle/pn2_oracle.c``.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product package
``regnet_for_3d_grasping_amd`` never does.

Shapes/dtypes follow the reference exactly: channel-first float32 inputs that may be
non-contiguous views, int64 indices, freshly allocated outputs.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpn2_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "pn2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


def _load(path=None):
    if path is None:
        build()
    lib = ctypes.CDLL(path or _SO)
    i64, f32, vp = ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
    lib.oracle_fps.argtypes = [vp, i64, i64, i64, vp]
    lib.oracle_ball_query.argtypes = [vp, vp, i64, i64, i64, f32, i64, vp, vp]
    lib.oracle_three_nn.argtypes = [vp, vp, i64, i64, i64, vp, vp]
    lib.oracle_group_fwd.argtypes = [vp, vp, i64, i64, i64, i64, i64, vp]
    lib.oracle_group_bwd.argtypes = [vp, vp, i64, i64, i64, i64, i64, vp]
    lib.oracle_interp_fwd.argtypes = [vp, vp, vp, i64, i64, i64, i64, vp]
    lib.oracle_interp_bwd.argtypes = [vp, vp, vp, i64, i64, i64, i64, vp]
    lib.oracle_radius_mask.argtypes = [vp, i64, vp, i64, i64, i64, f32, vp]
    for name in ("oracle_fps", "oracle_ball_query", "oracle_three_nn", "oracle_group_fwd",
                 "oracle_group_bwd", "oracle_interp_fwd", "oracle_interp_bwd", "oracle_radius_mask"):
        getattr(lib, name).restype = ctypes.c_int
    return lib


_lib = _load()


@__import__("contextlib").contextmanager
def fma_contracted():
    """Measurement only (scripts/fma_sensitivity.py): inside the block the FPS / ball-query / 3-NN distances are
    evaluated with nvcc-style FMA contraction (pn2_oracle.c: sqdist_cuda, -DORACLE_FMA_CONTRACT)."""
    global _lib
    so = os.path.join(_HERE, "_build", "libpn2_oracle_fma.so")
    src = os.path.join(_HERE, "pn2_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/libpn2_oracle_fma.so"])
    saved, _lib = _lib, _load(so)
    try:
        yield
    finally:
        _lib = saved


@__import__("contextlib").contextmanager
def native_build():
    """bench.py's cpu_baseline leg: inside the block the kernels come from a ``-O3 -march=native`` build of the same source
    made on THIS host (the portable -O2 library stays the checker's).  Falls back to the portable build when
    the host has no compiler."""
    global _lib
    import socket
    so = os.path.join(_HERE, "_build", "libpn2_oracle_native.so")
    stamp = so + ".host"
    src = os.path.join(_HERE, "pn2_oracle.c")
    host = socket.gethostname()
    try:
        fresh = (os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src) and os.path.exists(stamp)
                 and open(stamp).read() == host)
        if not fresh:
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "_build/libpn2_oracle_native.so"])
            with open(stamp, "w") as f:
                f.write(host)
        native = _load(so)
    except (OSError, subprocess.CalledProcessError):
        native = None
    saved = _lib
    if native is not None:
        _lib = native
    try:
        yield native is not None
    finally:
        _lib = saved


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("oracle %s failed (rc=%d)" % (what, rc))


def _cpu_f32(x, name):
    if x.is_cuda:
        raise RuntimeError("%s: the oracle works on CPU tensors" % name)
    if x.dtype != torch.float32:
        raise RuntimeError("%s must be float32" % name)
    return x


def _xyz_rows(x):
    """(B,3,N) view -> contiguous (B,N,3), as the reference's .transpose(1,2).contiguous()."""
    return x.transpose(1, 2).contiguous()


def farthest_point_sample(points, num_centroids):
    _cpu_f32(points, "points")
    if points.size(1) != 3:
        raise RuntimeError("points.size(1) does not equal to 3")
    B, _, N = points.shape
    if not num_centroids > 0:
        raise RuntimeError("num_centroids is not greater than 0")
    if not N >= num_centroids:
        raise RuntimeError("num_points is less than num_centroids")
    p = _xyz_rows(points)
    index = torch.zeros(B, num_centroids, dtype=torch.int64)
    _check(_lib.oracle_fps(p.data_ptr(), B, N, num_centroids, index.data_ptr()), "fps")
    return index


def ball_query(points, centroids, radius, num_neighbours):
    _cpu_f32(points, "points"); _cpu_f32(centroids, "centroids")
    if points.size(1) != 3 or centroids.size(1) != 3:
        raise RuntimeError("size(1) does not equal to 3")
    B, _, N1 = points.shape
    N2 = centroids.size(2)
    p, c = _xyz_rows(points), _xyz_rows(centroids)
    index = torch.zeros(B, N2, num_neighbours, dtype=torch.int64)
    count = torch.zeros(B, N2, dtype=torch.int64)
    _check(_lib.oracle_ball_query(p.data_ptr(), c.data_ptr(), B, N1, N2, float(radius),
                                  num_neighbours, index.data_ptr(), count.data_ptr()), "ball_query")
    return [index, count]


def point_search(query_xyz, key_xyz, num_neighbours):
    _cpu_f32(query_xyz, "query_xyz"); _cpu_f32(key_xyz, "key_xyz")
    B, _, N1 = query_xyz.shape
    N2 = key_xyz.size(2)
    if key_xyz.size(0) != B or query_xyz.size(1) != 3 or key_xyz.size(1) != 3:
        raise RuntimeError("point_search: shape mismatch")
    if num_neighbours != 3:
        raise RuntimeError("num_neighbours does not equal to K")
    if N2 < 3:
        raise RuntimeError("num_key is less than num_neighbours")
    q, k = _xyz_rows(query_xyz), _xyz_rows(key_xyz)
    index = torch.zeros(B, N1, 3, dtype=torch.int64)
    dist = torch.zeros(B, N1, 3, dtype=torch.float32)
    _check(_lib.oracle_three_nn(q.data_ptr(), k.data_ptr(), B, N1, N2, index.data_ptr(), dist.data_ptr()),
           "three_nn")
    return [index, dist]


def group_points_forward(input, index):
    _cpu_f32(input, "input")
    if input.dim() != 3 or index.dim() != 3 or index.size(0) != input.size(0):
        raise RuntimeError("group_points_forward: shape mismatch")
    B, C, N1 = input.shape
    _, N2, K = index.shape
    x, idx = input.contiguous(), index.contiguous()
    out = torch.zeros(B, C, N2, K, dtype=torch.float32)
    _check(_lib.oracle_group_fwd(x.data_ptr(), idx.data_ptr(), B, C, N1, N2, K, out.data_ptr()), "group_fwd")
    return out


def group_points_backward(grad_output, index, num_points):
    _cpu_f32(grad_output, "grad_output")
    B, C, N2, K = grad_output.shape
    if index.dim() != 3 or tuple(index.shape) != (B, N2, K):
        raise RuntimeError("group_points_backward: shape mismatch")
    g, idx = grad_output.contiguous(), index.contiguous()
    grad_in = torch.zeros(B, C, num_points, dtype=torch.float32)
    _check(_lib.oracle_group_bwd(g.data_ptr(), idx.data_ptr(), B, C, num_points, N2, K, grad_in.data_ptr()),
           "group_bwd")
    return grad_in


def interpolate_forward(input, index, weight):
    _cpu_f32(input, "input"); _cpu_f32(weight, "weight")
    B, C, M = input.shape
    N = index.size(1)
    if index.size(0) != B or index.size(2) != 3 or tuple(weight.shape) != (B, N, 3):
        raise RuntimeError("interpolate_forward: shape mismatch")
    x, idx, w = input.contiguous(), index.contiguous(), weight.contiguous()
    out = torch.zeros(B, C, N, dtype=torch.float32)
    _check(_lib.oracle_interp_fwd(x.data_ptr(), idx.data_ptr(), w.data_ptr(), B, C, M, N, out.data_ptr()),
           "interp_fwd")
    return out


def interpolate_backward(grad_output, index, weight, num_inst):
    _cpu_f32(grad_output, "grad_output"); _cpu_f32(weight, "weight")
    B, C, N = grad_output.shape
    if index.size(0) != B or index.size(2) != 3 or tuple(weight.shape) != (B, N, 3):
        raise RuntimeError("interpolate_backward: shape mismatch")
    g, idx, w = grad_output.contiguous(), index.contiguous(), weight.contiguous()
    grad_in = torch.zeros(B, C, num_inst, dtype=torch.float32)
    _check(_lib.oracle_interp_bwd(g.data_ptr(), idx.data_ptr(), w.data_ptr(), B, C, num_inst, N,
                                  grad_in.data_ptr()), "interp_bwd")
    return grad_in


# dgcnn_ext surface (functions/csrc/main.cpp:3-6): same gather / scatter-add.
def gather_knn_forward(input, index):
    return group_points_forward(input, index)


def gather_knn_backward(grad_output, index):
    return group_points_backward(grad_output, index, grad_output.size(2))


def radius_mask(points, centres, radius):
    """get_regiondataset.py:279-295: (NC,N) bool mask of sqrt(d2) <= R (test helper)."""
    p = points.contiguous().float()
    c = centres.contiguous().float()
    mask = torch.zeros(c.size(0), p.size(0), dtype=torch.uint8)
    _check(_lib.oracle_radius_mask(p.data_ptr(), p.stride(0), c.data_ptr(), c.stride(0), p.size(0), c.size(0),
                                   float(radius), mask.data_ptr()), "radius_mask")
    return mask.bool()
