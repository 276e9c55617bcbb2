"""Training-mode BatchNorm + ReLU (+ max over the K neighbours) of the shared-MLP blocks as fused HIP passes
(csrc/bn_train.hip), with autograd.

What it replaces, at the reference's operator granularity: ``nn.BatchNorm1d/2d`` followed by ``nn.ReLU(inplace=True)``
inside every Conv1d / Conv2d block (multi_model/utils/pn2_utils/nn/modules/conv.py:30-36, :70-76) and the
``torch.max(new_feature, 3)`` that ends a set-abstraction block (modules.py:245).  Same values up to fp32 rounding
(statistics are accumulated in fp64); running statistics and ``num_batches_tracked`` are updated exactly like torch.
GPU only -- there is no CPU path; callers keep torch's modules for anything ``supported`` rejects.
"""
import torch

from . import _lib

ENABLED = True
_check = _lib.check
_L = _lib.lib


def supported(bn, x, pool_group=0):
    """True when the fused kernels cover this BatchNorm module and input."""
    if not (ENABLED and bn.training and x.is_cuda and x.dtype == torch.float32 and x.dim() in (3, 4)):
        return False
    if not (bn.affine and bn.track_running_stats and bn.momentum is not None and bn.weight.dtype == torch.float32):
        return False
    if x.numel() == 0 or x.shape[0] > 65535 or x.shape[1] > 65535:
        return False
    if pool_group:
        return x.dim() == 4 and x.shape[3] == pool_group and 4 <= pool_group <= 256 and pool_group & (pool_group - 1) == 0
    return True


def _aligned(t):
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


class _BnReluTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu, pool_group, sums=None):
        x = _aligned(x)
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        dev = x.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            ws = sums if sums is not None else torch.empty((_L.regnet_bn_workspace_bytes(C),), dtype=torch.uint8, device=dev)
            mean = torch.empty((C,), dtype=torch.float32, device=dev)
            invstd = torch.empty((C,), dtype=torch.float32, device=dev)
            if pool_group:
                y = torch.empty(x.shape[:-1], dtype=torch.float32, device=dev)
                index = torch.empty(x.shape[:-1], dtype=torch.int32, device=dev)
            else:
                y, index = torch.empty_like(x), None
            gamma, beta = gamma.contiguous(), beta.contiguous()
            # sums: the convolution that produced x left the statistics (conv1x1_train FUSE_STATS): no pass over x for them
            fwd = _L.regnet_bn_relu_train_fwd_f32 if sums is None else _L.regnet_bn_relu_train_fwd_from_sums_f32
            _check(fwd(x.data_ptr(), B, C, L, gamma.data_ptr(), beta.data_ptr(), float(eps),
                       float(momentum), running_mean.data_ptr(), running_var.data_ptr(),
                       int(relu), int(pool_group), y.data_ptr(),
                       index.data_ptr() if index is not None else None, mean.data_ptr(),
                       invstd.data_ptr(), ws.data_ptr(), stream), "bn_relu_train_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, invstd, *((y, index) if pool_group else ()))
        ctx.relu, ctx.pool_group = int(relu), int(pool_group)
        return y

    @staticmethod
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        x, gamma, beta, mean, invstd = saved[:5]
        y, index = (saved[5], saved[6]) if ctx.pool_group else (None, None)
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        dev = x.device
        dy = _aligned(dy)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            ws = torch.empty((_L.regnet_bn_workspace_bytes(C),), dtype=torch.uint8, device=dev)
            dx = torch.empty_like(x)
            dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
            dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
            _check(_L.regnet_bn_relu_train_bwd_f32(x.data_ptr(), y.data_ptr() if y is not None else None, dy.data_ptr(),
                                                   index.data_ptr() if index is not None else None, B, C, L,
                                                   gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                   ctx.relu, ctx.pool_group, dx.data_ptr(), dgamma.data_ptr(),
                                                   dbeta.data_ptr(), ws.data_ptr(), stream), "bn_relu_train_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


class Pending:
    """A training BatchNorm (+ ReLU) whose statistics exist but whose output does not: ``x`` is its INPUT, ``scale`` / ``shift``
    the normalisation as a per-channel affine.  The convolution that consumes it applies the affine to its own operand
    (conv1x1_train.conv1x1_of_pending) and runs this BatchNorm's backward on its input gradient."""
    __slots__ = ("x", "gamma", "beta", "mean", "invstd", "scale", "shift", "relu")


def _sums_of(x):
    """Per-channel (sum, sum of squares) of ``x`` (2 C float64) if the convolution that produced this very tensor left them
    (conv1x1_train._with_sums), else None."""
    sums = getattr(x, "_bn_sums", None)
    if sums is None or sums.numel() != 2 * x.shape[1] or sums.dtype != torch.float64 or sums.device != x.device:
        return None
    return sums


def bn_stats(bn, x, relu=True):
    """Batch statistics of ``bn`` over ``x`` (running statistics and ``num_batches_tracked`` updated as the module's forward
    would) -> Pending; check ``supported`` first."""
    if not supported(bn, x):
        raise RuntimeError("bn_train.bn_stats: unsupported module / input (call supported() first)")
    bn.num_batches_tracked.add_(1)
    sums = _sums_of(x)
    xc = _aligned(x)
    B, C = xc.shape[0], xc.shape[1]
    L = xc.numel() // (B * C)
    dev = xc.device
    p = Pending()
    with torch.cuda.device(dev):
        out = torch.empty((4, C), dtype=torch.float32, device=dev)
        p.mean, p.invstd, p.scale, p.shift = out[0], out[1], out[2], out[3]
        p.gamma, p.beta = bn.weight, bn.bias
        g, b = bn.weight.detach().contiguous(), bn.bias.detach().contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        if sums is not None:      # left by the convolution that produced x (conv1x1_train FUSE_STATS): x is not read
            _check(_L.regnet_bn_train_stats_from_sums_f32(B, C, L, g.data_ptr(), b.data_ptr(), float(bn.eps), float(bn.momentum),
                                                          bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                                          p.mean.data_ptr(), p.invstd.data_ptr(), p.scale.data_ptr(),
                                                          p.shift.data_ptr(), sums.data_ptr(), stream), "bn_train_stats_from_sums")
        else:
            ws = torch.empty((_L.regnet_bn_workspace_bytes(C),), dtype=torch.uint8, device=dev)
            _check(_L.regnet_bn_train_stats_f32(xc.data_ptr(), B, C, L, g.data_ptr(), b.data_ptr(), float(bn.eps),
                                                float(bn.momentum), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                                p.mean.data_ptr(), p.invstd.data_ptr(), p.scale.data_ptr(), p.shift.data_ptr(),
                                                ws.data_ptr(), stream), "bn_train_stats")
    p.x, p.relu = xc, int(relu)
    return p


def bn_backward(x, dz, gamma, beta, mean, invstd, relu):
    """Training BatchNorm (+ ReLU) backward given the gradient ``dz`` of its output: -> (dx, dgamma, dbeta)."""
    B, C = x.shape[0], x.shape[1]
    L = x.numel() // (B * C)
    dev = x.device
    dz = _aligned(dz)
    gamma, beta = gamma.detach().contiguous(), beta.detach().contiguous()
    with torch.cuda.device(dev):
        ws = torch.empty((_L.regnet_bn_workspace_bytes(C),), dtype=torch.uint8, device=dev)
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        _check(_L.regnet_bn_relu_train_bwd_f32(x.data_ptr(), None, dz.data_ptr(), None, B, C, L, gamma.data_ptr(), beta.data_ptr(),
                                               mean.data_ptr(), invstd.data_ptr(), int(relu), 0, dx.data_ptr(), dgamma.data_ptr(),
                                               dbeta.data_ptr(), ws.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
               "bn_relu_train_bwd")
    return dx, dgamma, dbeta


def bn_relu(bn, x, relu=True, pool_group=0):
    """``[max over the last axis of] [relu](bn(x))`` for a BatchNorm module in training mode; check ``supported`` first."""
    if not supported(bn, x, pool_group):
        raise RuntimeError("bn_train.bn_relu: unsupported module / input (call supported() first)")
    bn.num_batches_tracked.add_(1)       # _BatchNorm.forward does this before F.batch_norm
    return _BnReluTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu,
                              pool_group, _sums_of(x))
