"""Fused training-mode BatchNorm + ReLU (+ max over neighbours) kernels (csrc/bn_train.hip) against torch's own modules
(the ops the reference uses: nn/modules/conv.py:30-36, modules.py:245), forward values, running statistics and every
gradient.  fp32 tolerance 1e-5 relative to the tensor's scale (statistics are fp64 inside the kernels, Welford fp32 in
torch)."""
import copy

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, tol=2e-5):
    a, b = a.detach(), b.detach()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, (err, scale)


def _close_most(a, b, tol=5e-5, outliers=1e-4):
    """Element-wise like _close, but a handful of elements may differ: a ReLU mask or an arg-max decided by a value within
    one ulp of the threshold / of the runner-up can legitimately fall the other way between two correct implementations,
    which moves one upstream gradient to another element."""
    a, b = a.detach(), b.detach()
    scale = max(1.0, float(b.abs().max()))
    bad = int(((a - b).abs() > tol * scale).sum())
    assert bad <= max(2, int(outliers * a.numel())), (bad, a.numel())


@pytest.mark.parametrize("shape,relu,pool", [
    ((2, 5, 37), True, 0), ((3, 16, 1024), True, 0), ((3, 16, 1024), False, 0), ((2, 8, 96, 64), True, 0),
    ((2, 8, 96, 64), True, 64), ((2, 7, 10, 4), True, 4), ((1, 3, 300, 16), False, 16), ((4, 130, 33, 64), True, 64),
    ((2, 4, 20000), True, 0), ((1, 2, 130, 256), True, 256),
])
def test_bn_relu_train_matches_torch(shape, relu, pool):
    from regnet_for_3d_grasping_amd import bn_train
    g = torch.Generator().manual_seed(sum(shape) + pool)
    x0 = (torch.randn(shape, generator=g) * 1.7 + 0.4)
    if pool:   # duplicated neighbours (ball-query padding) make exact ties
        x0[..., shape[-1] // 2:] = x0[..., :1]
    C = shape[1]
    ref = (nn.BatchNorm2d if len(shape) == 4 else nn.BatchNorm1d)(C).to(DEV)
    with torch.no_grad():
        ref.weight.copy_(torch.randn(C, generator=g))       # negative gammas too: the max must be taken AFTER the affine
        ref.bias.copy_(torch.randn(C, generator=g) * 0.3)
        ref.running_mean.copy_(torch.randn(C, generator=g))
        ref.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    mine = copy.deepcopy(ref)
    ref.train(), mine.train()
    xa = x0.to(DEV).requires_grad_(True)
    xb = x0.to(DEV).requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        ya = ref(xa)
    if relu:
        ya = torch.relu(ya)
    if pool:
        ya = torch.max(ya, 3)[0]
    assert bn_train.supported(mine, xb, pool)
    yb = bn_train.bn_relu(mine, xb, relu, pool)
    assert yb.shape == ya.shape
    _close(yb, ya)
    _close(mine.running_mean, ref.running_mean)
    _close(mine.running_var, ref.running_var)
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    up = torch.randn(ya.shape, generator=g).to(DEV)
    ya.backward(up)
    yb.backward(up)
    # which of several exactly tied maxima receives the gradient is unspecified (torch: any; here: the first), so with
    # pooling compare what the caller consumes: gradients summed over each group of identical rows -- and the parameters
    _close(mine.weight.grad, ref.weight.grad, 5e-5)
    _close(mine.bias.grad, ref.bias.grad, 5e-5)
    if pool:
        k = shape[-1] // 2
        ga = torch.cat([xa.grad[..., :1] + xa.grad[..., k:].sum(-1, keepdim=True), xa.grad[..., 1:k]], -1)
        gb = torch.cat([xb.grad[..., :1] + xb.grad[..., k:].sum(-1, keepdim=True), xb.grad[..., 1:k]], -1)
        _close_most(gb, ga)
        _close(xb.grad.sum(-1), xa.grad.sum(-1), 1e-4)     # invariant to WHICH neighbour received a maximum's gradient
    else:
        _close_most(xb.grad, xa.grad)


def test_shared_mlp_training_uses_the_fused_passes_and_matches_torch():
    """SharedMLP / set-abstraction reduction in training mode: fused path == torch path (same module, fused off)."""
    from regnet_for_3d_grasping_amd import bn_train
    from regnet_for_3d_grasping_amd.pn2_utils.nn import SharedMLP
    torch.manual_seed(5)
    a = SharedMLP(6, (32, 32, 64), ndim=2).to(DEV).train()
    b = copy.deepcopy(a)
    x = torch.randn(2, 6, 50, 64, device=DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = a(xa, pool_max=True)
    bn_train.ENABLED = False
    try:
        yb = b(xb, pool_max=True)
    finally:
        bn_train.ENABLED = True
    assert ya.shape == (2, 64, 50)
    _close(ya, yb)
    up = torch.randn_like(ya)
    ya.backward(up), yb.backward(up)
    _close_most(xa.grad, xb.grad, 1e-4)
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        _close(p.grad, q.grad, 1e-4)
    for (k, p), (_, q) in zip(a.named_buffers(), b.named_buffers()):
        _close(p.float(), q.float())


@pytest.mark.parametrize("shape,channels,pool,deferred", [
    ((2, 64, 300, 64), (64, 128, 128), True, 2),      # L = 19 200: ragged last point tile
    ((3, 128, 4112), (128, 64, 256), False, 2),       # 1-D, L % 256 = 16
    ((8, 32, 2048, 64), (64, 64), True, 1),           # many point tiles per workgroup
    ((1, 256, 640), (512, 256, 128), False, 2),       # fewer tiles than workgroup slots
    ((2, 6, 128, 64), (64, 64, 128), True, 2),        # first layer off the native kernels (6 channels): its BatchNorm still deferred
    ((2, 512, 1024), (1024, 256, 128), False, 1),     # second layer has 1 024 input channels: its BatchNorm the plain way
    ((2, 128, 1000), (128, 128), False, 0)])          # L % 16 != 0: nothing deferred
def test_deferred_batchnorm_between_convolutions(shape, channels, pool, deferred):
    """conv1x1_train.DEFER_BN: conv -> bn -> relu -> conv of a shared-MLP stack with the BatchNorm taken as statistics only
    and applied by the NEXT convolution to its operand fragments, forward and weight gradient (csrc/tgemm.hip B_AFFINE): the
    same outputs, running statistics and gradients as with the activation written out, and as torch's own modules in float64
    (nn/modules/conv.py:30-36, mlp.py:95-107)."""
    from regnet_for_3d_grasping_amd import conv1x1_train
    from regnet_for_3d_grasping_amd.pn2_utils.nn import SharedMLP
    torch.manual_seed(sum(shape))
    a = SharedMLP(shape[1], channels, ndim=len(shape) - 2).to(DEV).train()
    for blk in a:                                       # non-trivial affine parameters, some negative gammas
        blk.bn.weight.data.uniform_(-1.0, 1.5)
        blk.bn.bias.data.uniform_(-0.5, 0.5)
    b, c = copy.deepcopy(a), copy.deepcopy(a).double()
    x = torch.randn(shape, device=DEV) * 2 + 0.3
    xa, xb, xc = (x.clone().requires_grad_(True), x.clone().requires_grad_(True), x.double().requires_grad_(True))
    before = conv1x1_train.DEFERRED["layers"]
    ya = a(xa, pool_max=pool)
    up = torch.randn_like(ya)
    ya.backward(up)
    assert conv1x1_train.DEFERRED["layers"] - before == deferred
    old, conv1x1_train.DEFER_BN = conv1x1_train.DEFER_BN, False
    try:
        yb = b(xb, pool_max=pool)
        yb.backward(up)
    finally:
        conv1x1_train.DEFER_BN = old
    assert conv1x1_train.DEFERRED["layers"] - before == deferred
    yc = c(xc)                                          # torch's modules in float64
    if pool:
        yc = yc.max(-1)[0]
    yc.backward(up.double())
    _close(ya, yb, 2e-5), _close(ya, yc.float(), 1e-4)
    _close_most(xa.grad, xb.grad, 1e-4), _close_most(xa.grad, xc.grad.float(), 2e-4)
    for (k, p), (_, q), (_, r) in zip(a.named_parameters(), b.named_parameters(), c.named_parameters()):
        scale = max(1.0, float(r.grad.abs().max()))
        assert float((p.grad - q.grad).abs().max()) <= 3e-4 * scale, k          # (fp32 sums over up to 2.1 M points)
        assert float((p.grad.double() - r.grad).abs().max()) <= 3e-4 * scale, k
    for (k, p), (_, q), (_, r) in zip(a.named_buffers(), b.named_buffers(), c.named_buffers()):
        _close(p.float(), q.float(), 1e-5), _close(p.float(), r.float(), 1e-5)


@pytest.mark.parametrize("B,Co,Ci,L,affine", [
    (2, 128, 128, 19200, False),      # the level-1 block's second layer, scaled down
    (2, 256, 128, 19200, True),       # its third: two channel tiles, operand with its own BatchNorm + ReLU
    (3, 160, 64, 4112, True),         # a partial channel tile, L % 256 = 16
    (1, 64, 256, 1000, False),        # L % 16 != 0 (plain forward only), ragged last point tile
    (8, 128, 32, 2048 * 64, False)])  # many tiles per workgroup: the fp64 LDS table accumulates ~40 of them
def test_statistics_left_by_the_convolution(B, Co, Ci, L, affine):
    """conv1x1_train.FUSE_STATS (csrc/tgemm.hip STATS): the per-channel sum / sum of squares of a convolution's output, taken
    from its output tiles, against a float64 evaluation -- and the output itself unchanged."""
    from regnet_for_3d_grasping_amd import conv1x1_train as ct
    torch.manual_seed(B + Co + L)
    x = torch.randn((B, Ci, L), device=DEV) * 1.5 + 0.4
    w = torch.randn((Co, Ci), device=DEV) / Ci ** 0.5
    assert ct.stats_ok(Co, Ci, L, affine)
    sums = ct.new_sums(Co, DEV)
    if affine:
        scale, shift = torch.rand(Ci, device=DEV) + 0.5, torch.randn(Ci, device=DEV) * 0.3
        y = ct.native_fwd_bnrelu(x, w, scale, shift, 1, sums)
        y0 = ct.native_fwd_bnrelu(x, w, scale, shift, 1)
        a = torch.relu(x.double() * scale.double()[None, :, None] + shift.double()[None, :, None])
    else:
        y, y0, a = ct.native_fwd(x, w, sums), ct.native_fwd(x, w), x.double()
    assert torch.equal(y, y0)
    ref = torch.einsum("oi,bil->bol", w.double(), a)
    n = B * L
    s_ref, q_ref = ref.sum((0, 2)), (ref * ref).sum((0, 2))
    s, q = sums[0::2], sums[1::2]
    # the kernel sums ITS fp32 outputs: against the float64 sums of those the error is the summation's alone
    s_own, q_own = y.double().sum((0, 2)), (y.double() ** 2).sum((0, 2))
    assert float(((s - s_own).abs() / (q_own * n).sqrt()).max()) < 2e-7
    assert float(((q - q_own).abs() / q_own).max()) < 2e-7
    mean, var = s / n, q / n - (s / n) ** 2
    mean_ref, var_ref = s_ref / n, q_ref / n - (s_ref / n) ** 2
    assert float((mean - mean_ref).abs().max()) < 1e-5 and float(((var - var_ref).abs() / var_ref).max()) < 1e-4


@pytest.mark.parametrize("B,Co,Ci,L", [(2, 128, 6, 128 * 64), (8, 128, 6, 5120 * 64), (3, 64, 3, 1000), (1, 256, 8, 4)])
def test_statistics_of_a_handful_of_input_channels_come_from_the_input_moments(B, Co, Ci, L):
    """native_fwd_smallci(..., sums): y = W x with <= 8 input channels -- the per-channel sum / sum of squares of y from the first
    and second moments of x (csrc/tgemm.hip conv_smallci_kernel<.., STATS>), against float64; the output itself unchanged."""
    from regnet_for_3d_grasping_amd import conv1x1_train as ct
    torch.manual_seed(B + Co + L)
    x = torch.randn((B, Ci, L), device=DEV) * 0.7 + torch.linspace(-1.0, 2.0, Ci, device=DEV)[None, :, None]
    w = torch.randn((Co, Ci), device=DEV)
    sums = ct.new_sums(Co, DEV)
    y = ct.native_fwd_smallci(x, w, sums)
    assert torch.equal(y, ct.native_fwd_smallci(x, w))
    ref = torch.einsum("oi,bil->bol", w.double(), x.double())
    n = B * L
    s_ref, q_ref = ref.sum((0, 2)), (ref * ref).sum((0, 2))
    s, q = sums[0::2], sums[1::2]
    mean, var = s / n, q / n - (s / n) ** 2
    mean_ref, var_ref = s_ref / n, q_ref / n - (s_ref / n) ** 2
    assert float((mean - mean_ref).abs().max()) < 2e-6 * (1.0 + float(mean_ref.abs().max()))
    # (the moments are fp32 products: the variance is a difference of two of them)
    assert bool(((var - var_ref).abs() <= 2e-5 * var_ref + 2e-6 * q_ref / n).all())


def test_unsupported_inputs_are_rejected_not_silently_wrong():
    from regnet_for_3d_grasping_amd import bn_train
    bn = nn.BatchNorm1d(4).to(DEV).train()
    assert not bn_train.supported(bn, torch.zeros(8, 4, device=DEV))            # (N, C) rows: torch's kernels
    assert not bn_train.supported(bn.eval(), torch.zeros(2, 4, 8, device=DEV))  # eval mode: running statistics
    bn.train()
    assert not bn_train.supported(bn, torch.zeros(2, 4, 8, 24, device=DEV), 24)  # group not a power of two
    with pytest.raises(RuntimeError):
        bn_train.bn_relu(bn, torch.zeros(8, 4, device=DEV))
