"""Per-point (1x1) convolution of the shared-MLP blocks for TRAINING on the GPU, as plain GEMMs in the tensors' own
channel-first layout (nn/modules/conv.py:20-36, :60-76: ``nn.Conv1d/2d(kernel_size=1, bias=False)``).

Why not ``F.conv2d``: MIOpen computes the weight gradient of these layers with an NHWC implicit-GEMM kernel and therefore
transposes the (B,C,L) activation and gradient tensors (up to 1.3 GB each) to channels-last and back on every call --
measured 3.6 ms of ``batched_transpose`` + 5.4 ms of ``igemm_wrw`` per iteration at 4 scenes.  Both operands of
dW[o,i] = sum_{b,l} dY[b,o,l] X[b,i,l] are already K-major in memory, so it is a batched GEMM over chunks of the point
axis (split-K: the output is only C_out x C_in) followed by a small sum; forward and input gradient are one batched GEMM
each.  Same values up to fp32 summation order.
"""
import os

import torch

ENABLED = True     # module switches (bench.py --set conv1x1_train.ENABLED=0); nothing in the product reads the environment
# 1: forward / input gradient / weight gradient on this repo's own fp32-MFMA kernels (csrc/tgemm.hip) whenever the
# shape qualifies (channel counts multiples of 16, see regnet_conv1x1_train_supported); 0: rocBLAS batched GEMMs only
NATIVE = True
# forward / input gradient by the persistent ticket-driven kernel (csrc/tgemm.hip: tgemm_stream_kernel); 0: one workgroup per tile
STREAM = True
# conv -> bn -> relu -> conv inside a shared-MLP stack without the normalised activation: the first convolution's BatchNorm is
# evaluated as statistics only, the second convolution applies the normalisation + ReLU to its operand fragments (forward and
# weight gradient) and runs the BatchNorm's backward on its input gradient (csrc/tgemm.hip B_AFFINE, bn_train.Pending)
DEFER_BN = True
# input channel counts without a native tile (259, 515, 3: [xyz | feature] rows) are zero-padded to the next multiple of 16
# (at least 32) so that all three contractions stay on csrc/tgemm.hip (_Conv1x1Padded); 0: rocBLAS batched GEMMs for them
PAD_CHANNELS = True
# EXPERIMENT, off: forward / input gradient with fp32-faithful products on the bf16 matrix pipe (csrc/tsplit.hip: every
# operand as three bf16 pieces, six products, fp32 accumulation).  Never a default path; bench.py reports it under its own name.
SPLIT_PRODUCTS = False
# the statistics pass of the BatchNorm that follows a convolution taken from the convolution's output tiles while they are in
# registers (csrc/tgemm.hip STATS: per-channel sum / sum of squares in fp64), so that bn_train does not read the output for them
FUSE_STATS = True
DEFERRED = {"layers": 0}     # convolutions that consumed a pending BatchNorm since import (tests, bench)
STATS_FUSED = {"layers": 0}  # convolutions that left the following BatchNorm's statistics since import


def reserve_stream_slots(slots):
    """Leave ``slots`` of the persistent kernels' 2 x CUs workgroup slots empty from now on (csrc/tgemm.hip); returns the
    previous value.  ``train_step`` reserves some while the segmentation head's backward runs beside the region stage."""
    from . import _lib
    return int(_lib.lib.regnet_conv1x1_stream_reserve_slots(int(slots)))


def _split(transposed, w, x, B, Co, Ci, L, scale=None, shift=None, relu=0):
    """``regnet_conv1x1_split_f32``: x (B, K, L) -> (B, M, L) with (M, K) = (Co, Ci) forward, (Ci, Co) transposed."""
    from . import _lib
    out = torch.empty((B, Ci if transposed else Co, L), dtype=torch.float32, device=x.device)
    ws = torch.empty((_lib.lib.regnet_conv1x1_split_workspace_bytes(Co, Ci, transposed),), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib.regnet_conv1x1_split_f32(transposed, w.data_ptr(), x.data_ptr(), out.data_ptr(), B, Co, Ci, L,
                                                     scale.data_ptr() if scale is not None else None,
                                                     shift.data_ptr() if shift is not None else None, int(relu), ws.data_ptr(),
                                                     _stream(x)), "conv1x1_split")
    return out


def _split_ok(Co, Ci, L, affine=False):
    from . import _lib
    return SPLIT_PRODUCTS and bool(_lib.lib.regnet_conv1x1_split_supported(Co, Ci, L)) and (not affine or Ci <= 1024)


def _native_ok(B, Co, Ci, L, wgrad=False):
    if not NATIVE:
        return False
    from . import _lib
    return bool(_lib.lib.regnet_conv1x1_train_supported(Co, Ci, L)) and (not wgrad or L % 16 == 0)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def supported(conv, x):
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() in (3, 4) and conv.bias is None):
        return False
    one = (1,) * (x.dim() - 2)
    zero = (0,) * (x.dim() - 2)
    return (tuple(conv.kernel_size) == one and tuple(conv.stride) == one and tuple(conv.dilation) == one
            and tuple(conv.padding) == zero and conv.groups == 1 and x.numel() > 0)


# ---- thin wrappers of the native kernels (module-level names so bench.py --train can bracket them with events) --------
def stats_ok(Co, Ci, L, affine=False):
    """The forward of this shape can leave the statistics of the BatchNorm that follows it (``sums`` of native_fwd /
    native_fwd_bnrelu)."""
    if not (FUSE_STATS and NATIVE and STREAM) or _split_ok(Co, Ci, L, affine):
        return False
    from . import _lib
    return bool(_lib.lib.regnet_conv1x1_fwd_stats_supported(Co, Ci, L, int(affine)))


def new_sums(Co, device):
    STATS_FUSED["layers"] += 1
    return torch.empty((2 * Co,), dtype=torch.float64, device=device)


def native_fwd(x, w, sums=None):
    """x (B, Ci, L) contiguous, w (Co, Ci) contiguous -> Y (B, Co, L) on csrc/tgemm.hip.  ``sums`` (2 Co float64, see
    ``stats_ok``): filled with the per-channel sum and sum of squares of Y."""
    from . import _lib
    B, Ci, L = x.shape
    Co = w.shape[0]
    if sums is None and _split_ok(Co, Ci, L):
        return _split(0, w, x, B, Co, Ci, L)
    y = torch.empty((B, Co, L), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        if sums is not None:
            from . import fused
            _lib.check(_lib.lib.regnet_conv1x1_fwd_stats_stream_f32(w.data_ptr(), x.data_ptr(), y.data_ptr(), B, Co, Ci, L, None,
                                                                    None, 0, fused._tickets(x.device).data_ptr(),
                                                                    sums.data_ptr(), _stream(x)), "conv1x1_fwd_stats")
        elif STREAM:
            from . import fused
            _lib.check(_lib.lib.regnet_conv1x1_fwd_stream_f32(w.data_ptr(), x.data_ptr(), y.data_ptr(), B, Co, Ci, L,
                                                              fused._tickets(x.device).data_ptr(), _stream(x)),
                       "conv1x1_fwd_stream")
        else:
            _lib.check(_lib.lib.regnet_conv1x1_fwd_f32(w.data_ptr(), x.data_ptr(), y.data_ptr(), B, Co, Ci, L, _stream(x)),
                       "conv1x1_fwd")
    return y


def native_dgrad(w, dy):
    """w (Co, Ci), dy (B, Co, L), both contiguous -> dX (B, Ci, L)."""
    from . import _lib
    B, Co, L = dy.shape
    Ci = w.shape[1]
    if _split_ok(Co, Ci, L):
        return _split(1, w, dy, B, Co, Ci, L)
    dx = torch.empty((B, Ci, L), dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device):
        if STREAM:
            from . import fused
            _lib.check(_lib.lib.regnet_conv1x1_dgrad_stream_f32(w.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, Co, Ci, L,
                                                                fused._tickets(dy.device).data_ptr(), _stream(dy)),
                       "conv1x1_dgrad_stream")
        else:
            _lib.check(_lib.lib.regnet_conv1x1_dgrad_f32(w.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, Co, Ci, L,
                                                         _stream(dy)), "conv1x1_dgrad")
    return dx


def native_wgrad(dy, x):
    """dy (B, Co, L), x (B, Ci, L), both contiguous -> dW (Co, Ci); deterministic split over the point axis."""
    from . import _lib
    L_ = _lib.lib
    B, Co, L = dy.shape
    Ci = x.shape[1]
    dw = torch.empty((Co, Ci), dtype=torch.float32, device=x.device)
    ws_bytes = L_.regnet_conv1x1_wgrad_workspace_bytes(B, Co, Ci, L)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x.device) if ws_bytes else None
    with torch.cuda.device(x.device):
        _lib.check(L_.regnet_conv1x1_wgrad_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, Co, Ci, L,
                                               ws.data_ptr() if ws is not None else None, _stream(x)), "conv1x1_wgrad")
    return dw


def native_fwd_bnrelu(x, w, scale, shift, relu, sums=None):
    """Y = W . [relu](scale * x + shift) (per input channel), x (B, Ci, L) contiguous: the forward of a convolution that
    consumes a pending training BatchNorm (csrc/tgemm.hip: tgemm_stream_kernel<.., B_AFFINE>).  ``sums``: as native_fwd."""
    from . import _lib, fused
    B, Ci, L = x.shape
    Co = w.shape[0]
    if sums is None and _split_ok(Co, Ci, L, affine=True):
        return _split(0, w, x, B, Co, Ci, L, scale, shift, relu)
    y = torch.empty((B, Co, L), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        if sums is not None:
            _lib.check(_lib.lib.regnet_conv1x1_fwd_stats_stream_f32(w.data_ptr(), x.data_ptr(), y.data_ptr(), B, Co, Ci, L,
                                                                    scale.data_ptr(), shift.data_ptr(), relu,
                                                                    fused._tickets(x.device).data_ptr(), sums.data_ptr(),
                                                                    _stream(x)), "conv1x1_fwd_bnrelu_stats")
            return y
        _lib.check(_lib.lib.regnet_conv1x1_fwd_bnrelu_stream_f32(w.data_ptr(), x.data_ptr(), y.data_ptr(), B, Co, Ci, L,
                                                                 scale.data_ptr(), shift.data_ptr(), relu,
                                                                 fused._tickets(x.device).data_ptr(), _stream(x)),
                   "conv1x1_fwd_bnrelu")
    return y


def native_wgrad_bnrelu(dy, x, scale, shift, relu):
    """dW = sum_b dY[b] . ([relu](scale * x[b] + shift))^T: the weight gradient of that convolution."""
    from . import _lib
    L_ = _lib.lib
    B, Co, L = dy.shape
    Ci = x.shape[1]
    dw = torch.empty((Co, Ci), dtype=torch.float32, device=x.device)
    ws_bytes = L_.regnet_conv1x1_wgrad_workspace_bytes(B, Co, Ci, L)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x.device) if ws_bytes else None
    with torch.cuda.device(x.device):
        _lib.check(L_.regnet_conv1x1_wgrad_bnrelu_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, Co, Ci, L,
                                                      scale.data_ptr(), shift.data_ptr(), relu,
                                                      ws.data_ptr() if ws is not None else None, _stream(x)),
                   "conv1x1_wgrad_bnrelu")
    return dw


# what bench.py --train brackets: name -> meta(args)
TIMED_OPS = {
    "native_fwd_bnrelu": lambda x, w, scale, shift, relu, sums=None: "B%d Co%d Ci%d L%d flop%d" % (
        x.shape[0], w.shape[0], x.shape[1], x.shape[2], 2 * x.shape[0] * w.shape[0] * x.shape[1] * x.shape[2]),
    "native_wgrad_bnrelu": lambda dy, x, scale, shift, relu: "B%d Co%d Ci%d L%d flop%d" % (
        dy.shape[0], dy.shape[1], x.shape[1], x.shape[2], 2 * dy.shape[0] * dy.shape[1] * x.shape[1] * x.shape[2]),
    "native_fwd": lambda x, w, sums=None: "B%d Co%d Ci%d L%d flop%d" % (x.shape[0], w.shape[0], x.shape[1], x.shape[2],
                                                             2 * x.shape[0] * w.shape[0] * x.shape[1] * x.shape[2]),
    "native_dgrad": lambda w, dy: "B%d Co%d Ci%d L%d flop%d" % (dy.shape[0], w.shape[0], w.shape[1], dy.shape[2],
                                                                2 * dy.shape[0] * w.shape[0] * w.shape[1] * dy.shape[2]),
    "native_wgrad": lambda dy, x: "B%d Co%d Ci%d L%d flop%d" % (dy.shape[0], dy.shape[1], x.shape[1], x.shape[2],
                                                                2 * dy.shape[0] * dy.shape[1] * x.shape[1] * x.shape[2]),
}


def _chunks(L, tiles, B):
    """Number of point-axis chunks for the weight gradient: a power of two dividing L, enough (chunk x tile) GEMMs for
    ~1000 workgroups, chunks of at least 1024 points."""
    s = 1
    while L % (2 * s) == 0 and L // (2 * s) >= 1024 and s * tiles * B < 1024:
        s *= 2
    return s


def _smallci_ok(x, Ci, L):
    return NATIVE and Ci <= 8 and L % 4 == 0 and x.data_ptr() % 16 == 0


def native_fwd_smallci(x, w, sums=None):
    """x (B, Ci <= 8, L) contiguous, w (Co, Ci) contiguous -> Y (B, Co, L): a store stream (csrc/tgemm.hip: conv_smallci_kernel).
    ``sums`` (2 Co float64): filled with the per-channel sum / sum of squares of Y, from the input's moments."""
    from . import _lib
    B, Ci, L = x.shape
    Co = w.shape[0]
    y = torch.empty((B, Co, L), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        if sums is not None:
            ws = torch.empty((_lib.lib.regnet_conv1x1_smallci_stats_workspace_bytes(B, Ci, L) // 8,), dtype=torch.float64,
                             device=x.device)
            _lib.check(_lib.lib.regnet_conv1x1_fwd_smallci_stats_f32(w.data_ptr(), x.data_ptr(), y.data_ptr(), B, Co, Ci, L,
                                                                     ws.data_ptr(), sums.data_ptr(), _stream(x)),
                       "conv1x1_fwd_smallci_stats")
            return y
        _lib.check(_lib.lib.regnet_conv1x1_fwd_smallci_f32(w.data_ptr(), x.data_ptr(), y.data_ptr(), B, Co, Ci, L, _stream(x)),
                   "conv1x1_fwd_smallci")
    return y


def native_wgrad_smallci(dy, x):
    """dy (B, Co, L), x (B, Ci <= 8, L), both contiguous -> dW (Co, Ci): partial matrices per scene and slice
    (csrc/tgemm.hip: wgrad_smallci_kernel), summed in a fixed order."""
    from . import _lib
    L_ = _lib.lib
    B, Co, L = dy.shape
    Ci = x.shape[1]
    n = int(L_.regnet_conv1x1_wgrad_smallci_partials(B, Co, L))
    part = torch.empty((n, Co, Ci), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L_.regnet_conv1x1_wgrad_smallci_f32(dy.data_ptr(), x.data_ptr(), part.data_ptr(), B, Co, Ci, L, _stream(x)),
                   "conv1x1_wgrad_smallci")
    return part.sum(0) if n > 1 else part.view(Co, Ci)


class _ConvSmallCo(torch.autograd.Function):
    """y = W . x + bias for 1 <= Co <= 4 output channels (csrc/tgemm.hip: conv_smallco_*): x (B, Ci, L) contiguous, w (Co, Ci),
    bias (Co) or None."""

    @staticmethod
    def forward(ctx, x, w, bias):
        from . import _lib
        B, Ci, L = x.shape
        Co = w.shape[0]
        w = w.contiguous()
        y = torch.empty((B, Co, L), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib.regnet_conv1x1_smallco_f32(0, w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                           x.data_ptr(), y.data_ptr(), B, Co, Ci, L, _stream(x)), "conv1x1_smallco")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        B, Ci, L = x.shape
        Co = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib.regnet_conv1x1_smallco_f32(1, w.data_ptr(), None, dy.data_ptr(), dx.data_ptr(), B, Co, Ci, L,
                                                               _stream(x)), "conv1x1_smallco")
        if ctx.needs_input_grad[1]:
            dw = native_wgrad_smallci(x, dy).t().contiguous()       # (Ci, Co) partial sums with the roles swapped -> (Co, Ci)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2))
        return dx, dw, db


def small_co_ok(conv, x):
    """``conv`` is a kernel-size-1 Conv1d / Conv2d with at most 4 output channels (with or without bias) that the store-stream
    kernels take for the CUDA float32 input ``x``."""
    if not (ENABLED and NATIVE and x.is_cuda and x.dtype == torch.float32 and x.dim() in (3, 4) and conv.weight.shape[0] <= 4):
        return False
    one, zero = (1,) * (x.dim() - 2), (0,) * (x.dim() - 2)
    if not (tuple(conv.kernel_size) == one and tuple(conv.stride) == one and tuple(conv.dilation) == one
            and tuple(conv.padding) == zero and conv.groups == 1 and x.numel() > 0):
        return False
    L = x.numel() // (x.shape[0] * x.shape[1])
    return L % 4 == 0 and conv.weight.shape[1] <= 4096


def conv1x1_small_co(conv, x):
    """``conv(x)`` for such a layer; check ``small_co_ok`` first."""
    B, Ci = x.shape[0], x.shape[1]
    Co = conv.weight.shape[0]
    xc = x.contiguous().view(B, Ci, -1)
    if xc.data_ptr() % 16:
        xc = xc.clone()
    y = _ConvSmallCo.apply(xc, conv.weight.view(Co, Ci), conv.bias)
    return y.view(B, Co, *x.shape[2:])


def _padded_channels(B, Co, Ci, L):
    """Input channel count rounded up so that csrc/tgemm.hip takes the layer (259 -> 272, 515 -> 528, 3 -> 32: the first layers
    of the blocks that see [xyz | feature] rows), or 0 when padding does not help / is not worth it."""
    if not (NATIVE and PAD_CHANNELS) or Ci <= 8:     # (a handful of channels: native_fwd_smallci; their gradients are tiny GEMMs)
        return 0
    Cip = max(32, (Ci + 15) // 16 * 16)
    if Cip == Ci or Cip > 2 * Ci + 32 or not _native_ok(B, Co, Cip, L, wgrad=True):
        return 0
    return Cip


class _Conv1x1Padded(torch.autograd.Function):
    """_Conv1x1 for an input channel count csrc/tgemm.hip has no tile for: zero channels are appended to x and zero columns to
    w (exact: they add 0.0 to every sum), forward / input gradient / weight gradient then run on the native kernels instead of
    rocBLAS batched GEMMs (8 launches per layer and scene in the backward)."""

    @staticmethod
    def forward(ctx, x, w, Cip, sums=None):
        B, Ci, L = x.shape
        xp = torch.zeros((B, Cip, L), dtype=torch.float32, device=x.device)
        xp[:, :Ci].copy_(x)
        wp = torch.zeros((w.shape[0], Cip), dtype=torch.float32, device=x.device)
        wp[:, :Ci].copy_(w)
        ctx.save_for_backward(xp, wp)
        ctx.Ci = Ci
        return native_fwd(xp, wp, sums)

    @staticmethod
    def backward(ctx, dy):
        xp, wp = ctx.saved_tensors
        dy = dy.contiguous()
        dx = native_dgrad(wp, dy)[:, :ctx.Ci] if ctx.needs_input_grad[0] else None
        dw = native_wgrad(dy, xp)[:, :ctx.Ci] if ctx.needs_input_grad[1] else None
        return dx, dw, None, None


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, sums=None):
        """x (B, Ci, L) contiguous, w (Co, Ci) -> (B, Co, L); ``sums``: see native_fwd (the caller checked ``stats_ok``)."""
        ctx.save_for_backward(x, w)
        B, Ci, L = x.shape
        Co = w.shape[0]
        if _native_ok(B, Co, Ci, L):
            return native_fwd(x, w.contiguous(), sums)
        if _smallci_ok(x, Ci, L):
            return native_fwd_smallci(x, w.contiguous(), sums)
        # bmm on the expanded weight, NOT torch.matmul: for (2-D, 3-D) operands matmul folds the batch by transposing the
        # activation -- a full copy each way
        return torch.bmm(w.unsqueeze(0).expand(x.shape[0], -1, -1), x)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        B, Ci, L = x.shape
        Co = w.shape[0]
        if _native_ok(B, Co, Ci, L):
            dx = dw = None
            if ctx.needs_input_grad[0]:
                dx = native_dgrad(w.contiguous(), dy)
            if ctx.needs_input_grad[1] and _native_ok(B, Co, Ci, L, wgrad=True):
                dw = native_wgrad(dy, x)
            if dw is not None or not ctx.needs_input_grad[1]:
                return dx, dw, None
        else:
            dx = torch.bmm(w.t().unsqueeze(0).expand(B, -1, -1), dy) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and NATIVE and Ci <= 8 and L % 4 == 0 and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0:
            return dx, native_wgrad_smallci(dy, x), None
        if ctx.needs_input_grad[1]:
            S = _chunks(L, ((Co + 127) // 128) * ((Ci + 127) // 128), B)
            Ls = L // S
            part = torch.empty((B, S, Co, Ci), dtype=torch.float32, device=x.device)
            for b in range(B):
                xs = x[b].view(Ci, S, Ls).permute(1, 2, 0)       # (S, Ls, Ci): a strided view, no copy
                ds = dy[b].view(Co, S, Ls).transpose(0, 1)       # (S, Co, Ls)
                torch.bmm(ds, xs, out=part[b])
            dw = part.sum((0, 1)) if B * S > 1 else part.view(Co, Ci)
        return dx, dw, None


class _BnReluConv(torch.autograd.Function):
    """y = W . [relu](bn(x)) with bn in training mode, given bn's statistics (bn_train.Pending): one node for the BatchNorm's
    apply, the ReLU and the convolution; [relu](bn(x)) is never stored."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w, mean, invstd, scale, shift, relu, sums=None):
        w = w.contiguous()
        y = native_fwd_bnrelu(x, w, scale, shift, relu, sums)
        ctx.save_for_backward(x, gamma, beta, w, mean, invstd, scale, shift)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import bn_train
        x, gamma, beta, w, mean, invstd, scale, shift = ctx.saved_tensors
        dy = dy.contiguous()
        dz = native_dgrad(w, dy)
        dw = native_wgrad_bnrelu(dy, x, scale, shift, ctx.relu) if ctx.needs_input_grad[3] else None
        dx, dgamma, dbeta = bn_train.bn_backward(x, dz, gamma, beta, mean, invstd, ctx.relu)
        return dx, dgamma, dbeta, dw, None, None, None, None, None, None


def pending_ok(conv, x):
    """``conv`` can consume a bn_train.Pending whose BatchNorm input is ``x`` (shape of the activation)."""
    if not (DEFER_BN and STREAM and NATIVE and supported(conv, x)):
        return False
    from . import _lib
    B, Ci = x.shape[0], x.shape[1]
    return bool(_lib.lib.regnet_conv1x1_bnrelu_supported(conv.weight.shape[0], Ci, x.numel() // max(B * Ci, 1)))


def _with_sums(y, sums):
    """Hands the statistics a convolution left (``sums``) to the BatchNorm that consumes its output: an attribute of the very
    tensor object the block passes on (bn_train.bn_stats / bn_relu look for it)."""
    if sums is not None:
        y._bn_sums = sums
    return y


def conv1x1_of_pending(conv, pending, shape, stats=False):
    """``conv([relu](bn(x)))`` for a bn_train.Pending of that BatchNorm; ``shape``: of x as the stack sees it ((B,C,N) or
    (B,C,N,K)); check ``pending_ok`` first.  ``stats``: a training BatchNorm follows -- leave its statistics with the output."""
    B, Ci = shape[0], shape[1]
    Co = conv.weight.shape[0]
    DEFERRED["layers"] += 1
    x = pending.x.view(B, Ci, -1)
    sums = new_sums(Co, x.device) if stats and stats_ok(Co, Ci, x.shape[2], affine=True) else None
    y = _BnReluConv.apply(x, pending.gamma, pending.beta, conv.weight.view(Co, Ci), pending.mean,
                          pending.invstd, pending.scale, pending.shift, pending.relu, sums)
    return _with_sums(y.view(B, Co, *shape[2:]), sums)


def gemm_conv(x, w, sums=None):
    """x (B, Ci, L) contiguous, w (Co, Ci) -> (B, Co, L): the autograd GEMM convolution itself (``sums``: see native_fwd; only
    with a native, unpadded shape that ``stats_ok`` accepts)."""
    B, Ci, L = x.shape
    Cip = _padded_channels(B, w.shape[0], Ci, L) if x.is_cuda and not _native_ok(B, w.shape[0], Ci, L) else 0
    if Cip:
        return _Conv1x1Padded.apply(x, w, Cip, sums)
    return _Conv1x1.apply(x, w, sums)


def conv1x1(conv, x, stats=False):
    """``conv(x)`` for a bias-free kernel-size-1 Conv1d / Conv2d; check ``supported`` first.  ``stats``: a training BatchNorm
    follows -- leave its statistics with the output where the kernel can."""
    B, Ci = x.shape[0], x.shape[1]
    Co = conv.weight.shape[0]
    xf = x.contiguous().view(B, Ci, -1)
    L = xf.shape[2]
    sums = None
    if stats and x.is_cuda and FUSE_STATS:
        if _native_ok(B, Co, Ci, L):
            sums = new_sums(Co, x.device) if stats_ok(Co, Ci, L) else None
        elif _padded_channels(B, Co, Ci, L):
            sums = new_sums(Co, x.device) if stats_ok(Co, _padded_channels(B, Co, Ci, L), L) else None
        elif _smallci_ok(xf, Ci, L):
            sums = new_sums(Co, x.device)      # (a handful of input channels: from the input's moments, native_fwd_smallci)
    y = gemm_conv(xf, conv.weight.view(Co, Ci), sums)
    return _with_sums(y.view(B, Co, *x.shape[2:]), sums)
