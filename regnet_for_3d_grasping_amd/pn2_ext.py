"""``pn2_ext`` for MI355X: the reference's seven pybind entry points
(multi_model/utils/pn2_utils/csrc/main.cpp:6-14) on top of libregnet_hip.so.

Same names, argument order, shapes, dtypes and error behaviour as the CUDA extension:
inputs must be GPU tensors (the reference's CHECK_CUDA -- there is no CPU path), may be
non-contiguous views, are never modified; outputs are freshly allocated; failed checks raise
RuntimeError.  Kernels are enqueued on torch's current stream of the input's device.
"""
import torch

from . import _lib

_check = _lib.check
_L = _lib.lib

# Source clouds with at least this many points are binned into a uniform grid first (csrc/grid.hip);
# smaller ones are cheaper to scan exhaustively from LDS.  Results are identical either way.
GRID_MIN_POINTS = 2048        # 3-NN keys
GRID_MIN_POINTS_BALL = 8192   # ball-query cloud (measured break-even between 5 120 and 25 600 points)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)  # CHECK_CUDA (sampling_kernel.cu:11)


def _need_f32(t, name):
    _need_gpu(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (REGNet's path is fp32 only)" % name)


def _need_i64(t, name):
    _need_gpu(t, name)
    if t.dtype != torch.int64:
        raise RuntimeError("%s must be int64" % name)


def _eq(a, b, text):
    if a != b:
        raise RuntimeError("%s" % text)  # CHECK_EQ


class FpsChain:
    """Optional third argument of ``farthest_point_sample`` for callers that sample a cloud which is itself a
    furthest-point-sampling sequence (PointNet++ levels 2 and 3 sample the previous level's centroids in pick order):
    ``prefix_ok`` in -- the ``first_tie`` tensor of the run that produced the cloud's order, or None -- and ``first_tie``
    out, (B,) int32 (include/regnet_hip.h: regnet_fps_chain_f32).  Scenes whose producing run had no tie among its first M
    picks get 0 .. M-1 without sampling; the result is the same tensor either way."""

    def __init__(self, prefix_ok=None):
        self.prefix_ok, self.first_tie = prefix_ok, None


def farthest_point_sample(points, num_centroids, chain=None):
    """points (B,3,N1) -> index (B,N2) int64.  csrc/sampling_kernel.cu:126-170.  ``chain``: see FpsChain (not part of the
    reference's signature; the result does not depend on it)."""
    _need_f32(points, "points")
    _eq(points.dim(), 3, "points must be (B, 3, N)")
    _eq(points.size(1), 3, "points.size(1) does not equal to 3")
    B, _, N = points.shape
    M = int(num_centroids)
    if not M > 0:
        raise RuntimeError("num_centroids is not greater than 0")
    if not N >= M:
        raise RuntimeError("num_points is less than num_centroids")
    with torch.cuda.device(points.device):
        index = torch.empty((B, M), dtype=torch.int64, device=points.device)
        ws_bytes = _L.regnet_fps_workspace_bytes(B, N, M)
        ws = torch.empty((ws_bytes // 4,), dtype=torch.float32, device=points.device) if ws_bytes else None
        sb, sc, sn = points.stride()
        if chain is None:
            _check(_L.regnet_fps_f32(points.data_ptr(), sb, sc, sn, B, N, M, index.data_ptr(),
                                     ws.data_ptr() if ws is not None else None, _stream(points)),
                   "farthest_point_sample")
        else:
            prefix = chain.prefix_ok
            if prefix is not None and (prefix.dtype != torch.int32 or prefix.numel() != B or not prefix.is_cuda):
                raise RuntimeError("FpsChain.prefix_ok must be a (B,) int32 GPU tensor")
            chain.first_tie = torch.empty((B,), dtype=torch.int32, device=points.device)
            _check(_L.regnet_fps_chain_f32(points.data_ptr(), sb, sc, sn, B, N, M, index.data_ptr(),
                                           ws.data_ptr() if ws is not None else None,
                                           prefix.contiguous().data_ptr() if prefix is not None else None,
                                           chain.first_tie.data_ptr(), _stream(points)), "farthest_point_sample")
        status_at = _L.regnet_fps_status_offset_bytes(B, N, M)
        if status_at >= 0:
            # cooperative sampling (N > 25 600): accumulate the launch's status word into the device's flag -- one tiny
            # launch on the same stream, no synchronisation; raise_if_fps_failed() reads it where the caller synchronises
            flag = _fps_flag(points.device)
            torch.bitwise_or(flag, ws[status_at // 4: status_at // 4 + 1].view(torch.int32), out=flag)
    return index


def gather_points(points, index, channels_last=False):
    """points (B,C,N) float32, index (B,M) int64 -> (B,C,M): ``points[b, :, index[b, m]]`` (pn2_utils/function.py:11-26's
    ``torch.gather``) as one native launch; any strides.  ``channels_last``: the result as a contiguous (B,M,C) tensor (rows of
    a (B,N,C) cloud handed in as its transposed view).  An index outside [0, N) flags the device's status word
    (``raise_if_fps_failed`` reports it where the caller synchronises) and yields 0."""
    _need_f32(points, "points")
    _need_i64(index, "index")
    _eq(points.dim(), 3, "points must be (B, C, N)")
    _eq(index.dim(), 2, "index must be (B, M)")
    _eq(points.size(0), index.size(0), "points and index differ in batch size")
    B, C, N = points.shape
    M = index.size(1)
    with torch.cuda.device(points.device):
        if channels_last:
            out = torch.empty((B, M, C), dtype=torch.float32, device=points.device)
            ob, om, oc = out.stride()
        else:
            out = torch.empty((B, C, M), dtype=torch.float32, device=points.device)
            ob, oc, om = out.stride()
        _check(_L.regnet_gather_points_f32(points.data_ptr(), *points.stride(), B, C, N, index.data_ptr(), *index.stride(), M,
                                           out.data_ptr(), ob, oc, om, _fps_flag(points.device).data_ptr(),
                                           _stream(points)), "gather_points")
    return out


def class_order(count):
    """count (...) int64 GPU tensor of ball-query member counts -> (count.numel(),) int64: the stable sort permutation by cost
    class (count > 32) + (count > 48) (``torch.argsort(cls, stable=True)``), one launch (regnet_class_order_i64)."""
    _need_i64(count, "count")
    flat = count.reshape(-1)
    flat = flat if flat.is_contiguous() else flat.contiguous()
    with torch.cuda.device(count.device):
        order = torch.empty((flat.numel(),), dtype=torch.int64, device=count.device)
        _check(_L.regnet_class_order_i64(flat.data_ptr(), flat.numel(), order.data_ptr(), _stream(count)), "class_order")
    return order


_fps_flags = {}


def _fps_flag(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _fps_flags:
        _fps_flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _fps_flags[key]


def raise_if_fps_failed():
    """RuntimeError if a cooperative furthest-point-sampling launch since the last check lost a partner workgroup (a
    scene's 2-4 workgroups exchange records every round and must all be resident; a poll gives up after 2^22 tries, flags
    the launch and stops instead of sampling on with diverged selections).  Synchronises; called where the caller
    synchronises anyway (end of pipeline.forward_scenes / ForwardPipeline.run), like region_ops.raise_if_out_of_range."""
    for flag in _fps_flags.values():
        if int(flag.item()):
            flag.zero_()
            raise RuntimeError("farthest_point_sample: a cooperating workgroup lost its partner (launch not fully "
                               "resident?) -- the returned indices are incomplete -- or gather_points saw an index "
                               "outside the cloud")


def ball_query(points, centroids, radius, num_neighbours):
    """points (B,3,N1), centroids (B,3,N2) -> [index (B,N2,K) int64, count (B,N2) int64].
    csrc/ball_query_kernel.cu:87-131."""
    _need_f32(points, "points")
    _need_f32(centroids, "centroids")
    _eq(points.size(1), 3, "points.size(1) does not equal to 3")
    _eq(centroids.size(1), 3, "centroids.size(1) does not equal to 3")
    _eq(centroids.size(0), points.size(0), "centroids.size(0) does not equal to batch_size")
    B, _, N1 = points.shape
    N2 = centroids.size(2)
    K = int(num_neighbours)
    with torch.cuda.device(points.device):
        index = torch.empty((B, N2, K), dtype=torch.int64, device=points.device)
        count = torch.empty((B, N2), dtype=torch.int64, device=points.device)
        if N1 >= GRID_MIN_POINTS_BALL and K <= 64 and B > 0 and float(radius) > 0:
            ws = torch.empty((_L.regnet_grid_workspace_bytes(B, N1),), dtype=torch.uint8, device=points.device)
            _check(_L.regnet_ball_query_grid_f32(points.data_ptr(), *points.stride(), centroids.data_ptr(),
                                                 *centroids.stride(), B, N1, N2, float(radius), K,
                                                 index.data_ptr(), count.data_ptr(), ws.data_ptr(),
                                                 _stream(points)), "ball_query")
        else:
            _check(_L.regnet_ball_query_f32(points.data_ptr(), *points.stride(), centroids.data_ptr(),
                                            *centroids.stride(), B, N1, N2, float(radius), K, index.data_ptr(),
                                            count.data_ptr(), _stream(points)), "ball_query")
    return [index, count]


def group_points_forward(input, index):
    """input (B,C,N1), index (B,N2,K) -> (B,C,N2,K).  csrc/grouping_kernel.cu:29-51."""
    _need_f32(input, "input")
    _need_i64(index, "index")
    _eq(input.dim(), 3, "input.dim() does not equal to 3")
    _eq(index.dim(), 3, "index.dim() does not equal to 3")
    _eq(index.size(0), input.size(0), "index.size(0) does not equal to batch_size")
    B, C, N1 = input.shape
    _, N2, K = index.shape
    with torch.cuda.device(input.device):
        idx = index.contiguous()
        out = torch.empty((B, C, N2, K), dtype=torch.float32, device=input.device)
        _check(_L.regnet_group_points_fwd_f32(input.data_ptr(), *input.stride(), idx.data_ptr(), B, C, N1, N2, K,
                                              out.data_ptr(), _stream(input)), "group_points_forward")
    return out


def group_points_backward(grad_output, index, num_points):
    """grad_output (B,C,N2,K), index (B,N2,K) -> grad_input (B,C,N1).  csrc/grouping_kernel.cu:103-149."""
    _need_f32(grad_output, "grad_output")
    _need_i64(index, "index")
    _eq(grad_output.dim(), 4, "grad_output.dim() does not equal to 4")
    _eq(index.dim(), 3, "index.dim() does not equal to 3")
    B, C, N2, K = grad_output.shape
    _eq(index.size(0), B, "index.size(0) does not equal to batch_size")
    _eq(index.size(1), N2, "index.size(1) does not equal to num_select")
    _eq(index.size(2), K, "index.size(2) does not equal to k")
    N1 = int(num_points)
    with torch.cuda.device(grad_output.device):
        idx = index.contiguous()
        grad_in = torch.empty((B, C, N1), dtype=torch.float32, device=grad_output.device)
        _check(_L.regnet_group_points_bwd_f32(grad_output.data_ptr(), *grad_output.stride(), idx.data_ptr(), B, C,
                                              N1, N2, K, grad_in.data_ptr(), _stream(grad_output)),
               "group_points_backward")
    return grad_in


def point_search(query_xyz, key_xyz, num_neighbours):
    """query (B,3,N1), key (B,3,N2) -> [index (B,N1,3) int64, squared distance (B,N1,3)].
    csrc/interpolate_kernel.cu:88-128."""
    _need_f32(query_xyz, "query_xyz")
    _need_f32(key_xyz, "key_xyz")
    B, _, N1 = query_xyz.shape
    N2 = key_xyz.size(2)
    _eq(key_xyz.size(0), B, "key_xyz.size(0) does not equal to batch_size")
    _eq(query_xyz.size(1), 3, "query_xyz.size(1) does not equal to 3")
    _eq(key_xyz.size(1), 3, "key_xyz.size(1) does not equal to 3")
    _eq(int(num_neighbours), 3, "num_neighbours does not equal to K")
    if not N2 >= 3:
        raise RuntimeError("num_key is less than num_neighbours")
    with torch.cuda.device(query_xyz.device):
        index = torch.empty((B, N1, 3), dtype=torch.int64, device=query_xyz.device)
        dist = torch.empty((B, N1, 3), dtype=torch.float32, device=query_xyz.device)
        if N2 >= GRID_MIN_POINTS and B > 0:
            ws = torch.empty((_L.regnet_grid_workspace_bytes(B, N2),), dtype=torch.uint8, device=query_xyz.device)
            _check(_L.regnet_three_nn_grid_f32(query_xyz.data_ptr(), *query_xyz.stride(), key_xyz.data_ptr(),
                                               *key_xyz.stride(), B, N1, N2, index.data_ptr(), dist.data_ptr(),
                                               ws.data_ptr(), _stream(query_xyz)), "point_search")
        else:
            _check(_L.regnet_three_nn_f32(query_xyz.data_ptr(), *query_xyz.stride(), key_xyz.data_ptr(),
                                          *key_xyz.stride(), B, N1, N2, index.data_ptr(), dist.data_ptr(),
                                          _stream(query_xyz)), "point_search")
    return [index, dist]


def _check_interp(first, index, weight, B, N):
    _need_i64(index, "index")
    _need_f32(weight, "weight")
    _eq(index.size(0), B, "index.size(0) does not equal to batch_size")
    _eq(index.size(2), 3, "index.size(2) does not equal to K")
    _eq(weight.size(0), B, "weight.size(0) does not equal to batch_size")
    _eq(weight.size(1), N, "weight.size(1) does not equal to num_select")
    _eq(weight.size(2), 3, "weight.size(2) does not equal to K")


def interpolate_forward(input, index, weight):
    """input (B,C,M), index/weight (B,N,3) -> (B,C,N).  csrc/interpolate_kernel.cu:187-232."""
    _need_f32(input, "input")
    B, C, M = input.shape
    N = index.size(1)
    _check_interp(input, index, weight, B, N)
    with torch.cuda.device(input.device):
        idx, w = index.contiguous(), weight.contiguous()
        out = torch.empty((B, C, N), dtype=torch.float32, device=input.device)
        _check(_L.regnet_interpolate_fwd_f32(input.data_ptr(), *input.stride(), idx.data_ptr(), w.data_ptr(), B, C,
                                             M, N, out.data_ptr(), _stream(input)), "interpolate_forward")
    return out


def interpolate_backward(grad_output, index, weight, num_inst):
    """grad_output (B,C,N) -> grad_input (B,C,M).  csrc/interpolate_kernel.cu:292-337."""
    _need_f32(grad_output, "grad_output")
    B, C, N = grad_output.shape
    _check_interp(grad_output, index, weight, B, N)
    M = int(num_inst)
    with torch.cuda.device(grad_output.device):
        idx, w = index.contiguous(), weight.contiguous()
        grad_in = torch.empty((B, C, M), dtype=torch.float32, device=grad_output.device)
        _check(_L.regnet_interpolate_bwd_f32(grad_output.data_ptr(), *grad_output.stride(), idx.data_ptr(),
                                             w.data_ptr(), B, C, M, N, grad_in.data_ptr(), _stream(grad_output)),
               "interpolate_backward")
    return grad_in
