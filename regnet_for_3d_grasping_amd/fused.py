"""Fused eval-mode forward of the set-abstraction / feature-propagation / head blocks on MI355X.

``pn2_utils.modules`` and ``pointnet2`` dispatch here when ``usable()`` holds (module in eval
mode, autograd off, GPU tensors).  The module API is unchanged -- channel-first ``(B,C,N)``
tensors in and out -- but internally activations are channels-last buffers driven through the
fp32-MFMA shared-MLP kernels of csrc/mlp.hip, and the tensors handed back are transposed *views*
of those buffers (so the next fused block reads them without a copy).

Per block (reference pn2_utils/modules.py:210-246 and :500-509, pointnet2.py:116-119):
  SA  : FPS -> ball query -> [gather(feat | xyz - centre) . W1] -> [. W2] -> [. W3 + max over K]
  FP  : 3-NN -> interpolate+concat (channels-last) -> [. W1] -> [. W2] (-> [. W3])
  head: [. W]*4 -> conv_score + bn_score + sigmoid
Eval-mode BatchNorm is folded into a per-channel (scale, shift) applied in the GEMM epilogue.
"""
import torch

from . import _lib, pn2_ext

ENABLED = True
CHAIN3 = True   # level-1 set-abstraction block as one register-chained kernel (see sa_features)
SPLITK_MAX_ROWS = 1024   # at most this many rows: the GEMM is "skinny" and is split along K (see mlp_layer)
PREMUL = True   # evaluate the first layer of wide set-abstraction blocks per source point (see sa_features)
PREMUL_CENTRE = True   # ... on mean-centred coordinates (a module switch like the others: bench.py --set fused.PREMUL_CENTRE=0)

_check = _lib.check
_L = _lib.lib


def usable(module, xyz):
    return ENABLED and xyz.is_cuda and not module.training and not torch.is_grad_enabled()


def supports_sa(module, feature):
    """True when the fused set-abstraction chain covers this module's configuration; otherwise the caller takes the
    operator-granular GPU path (pn2_ext group / torch conv), exactly like the reference."""
    grouper = getattr(module, "grouper", None)
    return (grouper is not None and grouper.num_neighbours == 64 and len(module.mlp) >= 2
            and (feature is None or (module.use_xyz and feature.dtype == torch.float32)))


def supports_fp(module, sparse_feature):
    return sparse_feature.dtype == torch.float32 and getattr(module.interpolator, "num_neighbors", 0) == 3


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _on_tensor_device(fn):
    """Run a kernel wrapper with the HIP device of its first GPU tensor argument current (hipLaunchKernel on a stream
    of another device fails with an invalid-resource-handle error): a model living on cuda:k must work whatever
    ``torch.cuda.current_device()`` is, like pn2_ext / region_ops / bn_train.  Free when the device is already
    current (the common case: one process per GPU)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapped


def _round_up(x, m):
    return (x + m - 1) // m * m


_ticket_pools = {}


def _tickets(device, n=1):
    """``n`` zeroed int32 work-queue heads for a ticket-driven chain kernel.  Carved out of a pool that is zeroed once per 256
    words ON THE STREAM THAT USES IT (a word is handed out once and never reused): one fill kernel per ~60 feature stages
    instead of one -- and its ~10 us launch gap -- in front of every chain launch."""
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture (pipeline.StageGraphs): the word is reused by every replay, so its zero-fill is a node of
        # the graph (memory from the graph's private pool)
        return torch.zeros((n,), dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream.cuda_stream)
    pool = _ticket_pools.get(key)
    if pool is None or pool[1] + n > pool[0].numel():
        pool = [torch.zeros((256,), dtype=torch.int32, device=device), 0]
        _ticket_pools[key] = pool
    out = pool[0][pool[1]:pool[1] + n]
    pool[1] += n
    return out


class _Layer:
    """One conv(1x1, bias-free or biased) + eval BatchNorm (+ReLU) packed for the GEMM kernel."""

    __slots__ = ("W", "W8", "scale", "shift", "N", "K", "Kpad", "relu", "premul")


def _pack(conv, bn, relu, col_order=None):
    w = conv.weight.detach().reshape(conv.weight.shape[0], -1).float()  # (N, K)
    N, K = w.shape
    if col_order is not None:
        w = w[:, col_order]
        K = w.shape[1]
    L = _Layer()
    L.N, L.K, L.Kpad, L.relu = N, K, _round_up(max(K, 1), 16), 1 if relu else 0
    Wp = torch.zeros((_round_up(N, 128), L.Kpad), dtype=torch.float32, device=w.device)
    Wp[:N, :K] = w
    if bn is not None:
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    else:
        scale = torch.ones(N, dtype=torch.float32, device=w.device)
        shift = torch.zeros(N, dtype=torch.float32, device=w.device)
    if conv.bias is not None:
        shift = shift + conv.bias.detach().float() * scale
    L.W, L.scale, L.shift = Wp.contiguous(), scale.contiguous(), shift.contiguous()
    L.W8 = None
    L.premul = None
    if K <= 8:  # narrow first layer: also keep the [N][8] form consumed by the layer-1-fused gather kernel
        W8 = torch.zeros((N, 8), dtype=torch.float32, device=w.device)
        W8[:, :K] = w
        L.W8 = W8.contiguous()
    return L


def _signature(module):
    """(address, version) of every parameter and buffer: changes with any in-place update (optimizer step, load_state_dict,
    .to()).  The LIST of tensors is cached on the module -- walking ``parameters()`` / ``buffers()`` is ~15 us per call and a
    forward asks ~30 times -- and re-read by ``prepack`` (every ``ForwardPipeline.run``) or ``forget``: replacing a Parameter
    OBJECT or a sub-module of a network that has already run needs one of the two."""
    tensors = module.__dict__.get("_regnet_sig_tensors")
    if tensors is None:
        tensors = list(module.parameters()) + list(module.buffers())
        module.__dict__["_regnet_sig_tensors"] = tensors
    return tuple((t.data_ptr(), t._version) for t in tensors)


def forget(*nets):
    """Drop the cached tensor lists of ``_signature`` below these networks (after structural surgery on a network)."""
    for net in nets:
        if net is not None:
            for m in net.modules():
                m.__dict__.pop("_regnet_sig_tensors", None)


def _packed_stack(owner, stack, first_col_order=None):
    """Packed layers of a SharedMLP (list of conv/bn/relu blocks), cached on ``owner`` and rebuilt
    whenever a parameter or BN buffer is modified (in-place updates bump ``_version``).
    ``first_col_order``: optional callable returning the column permutation of the first layer (only
    evaluated on a cache miss -- it involves a host->device copy, which would stall the stream)."""
    sig = _signature(stack)
    cache = getattr(owner, "_regnet_packed", None)
    if cache is None or cache[0] != sig:
        layers = []
        order = first_col_order() if first_col_order is not None else None
        for i, block in enumerate(stack):
            layers.append(_pack(block.conv, block.bn, block.relu is not None, order if i == 0 else None))
        cache = (sig, layers)
        owner._regnet_packed = cache
    return cache[1]


# ---- thin kernel wrappers (module-level names so bench.py can bracket them with events) ------
@_on_tensor_device
def mlp_layer(A, Ka, layer, P, pool_group=0):
    """A: channels-last (P, lda) float32 buffer whose first Ka columns are valid."""
    rows = P // pool_group if pool_group else P
    out = torch.empty((rows, layer.N), dtype=torch.float32, device=A.device)
    if not pool_group and P <= SPLITK_MAX_ROWS and layer.Kpad >= 128:
        # skinny problem (the region heads: B*64 rows): a 128-row tile grid cannot fill the chip, so cut K into
        # slices of >= 64 that run on different workgroups (deterministic: added in order by a second kernel)
        tiles = ((P + 127) // 128) * ((layer.N + 127) // 128)
        ksplit = max(1, min(layer.Kpad // 64, 256 // max(tiles, 1)))
        if ksplit > 1:
            ws = torch.empty((_L.regnet_mlp_splitk_workspace_bytes(P, layer.N, ksplit),), dtype=torch.uint8,
                             device=A.device)
            _check(_L.regnet_mlp_layer_splitk_f32(A.data_ptr(), A.stride(0), Ka, layer.W.data_ptr(), layer.Kpad,
                                                  layer.scale.data_ptr(), layer.shift.data_ptr(), out.data_ptr(),
                                                  out.stride(0), P, layer.N, layer.relu, ksplit, ws.data_ptr(),
                                                  _stream(A)), "mlp_layer_splitk")
            return out
    _check(_L.regnet_mlp_layer_f32(A.data_ptr(), A.stride(0), Ka, layer.W.data_ptr(), layer.Kpad,
                                   layer.scale.data_ptr(), layer.shift.data_ptr(), out.data_ptr(), out.stride(0), P,
                                   layer.N, layer.relu, pool_group, _stream(A)), "mlp_layer")
    return out


@_on_tensor_device
def sa_layer1(feature, xyz, nbr, ctr, layer, B, M, group):
    """Gather-fused first SA layer.  feature (B,Cf,N) any strides or None; xyz (B,3,N) any strides."""
    out = torch.empty((B * M * group, layer.N), dtype=torch.float32, device=xyz.device)
    if feature is None:
        fptr, fb, fn, fc, Cf = None, 0, 0, 0, 0
    else:
        fptr, (fb, fc, fn), Cf = feature.data_ptr(), feature.stride(), feature.size(1)
    _check(_L.regnet_sa_layer1_f32(fptr, fb, fn, fc, Cf, xyz.data_ptr(), *xyz.stride(), nbr.data_ptr(), ctr.data_ptr(),
                                   B, M, group, layer.W.data_ptr(), layer.Kpad, layer.scale.data_ptr(),
                                   layer.shift.data_ptr(), out.data_ptr(), out.stride(0), layer.N, layer.relu,
                                   _stream(xyz)), "sa_layer1")
    return out


@_on_tensor_device
def sa_layer12(feature, xyz, nbr, ctr, first, layer, B, M, group, pool_group=0):
    """Gather + layer 1 (VALU, inside the operand load) + layer 2 (MFMA) of a narrow-input SA block."""
    P = B * M * group
    rows = P // pool_group if pool_group else P
    out = torch.empty((rows, layer.N), dtype=torch.float32, device=xyz.device)
    if feature is None:
        fptr, fb, fn, fc, Cf = None, 0, 0, 0, 0
    else:
        fptr, (fb, fc, fn), Cf = feature.data_ptr(), feature.stride(), feature.size(1)
    _check(_L.regnet_sa_layer12_f32(fptr, fb, fn, fc, Cf, xyz.data_ptr(), *xyz.stride(), nbr.data_ptr(),
                                    ctr.data_ptr(), B, M, group, first.W8.data_ptr(), first.scale.data_ptr(),
                                    first.shift.data_ptr(), first.N, layer.W.data_ptr(), layer.Kpad,
                                    layer.scale.data_ptr(), layer.shift.data_ptr(), out.data_ptr(), out.stride(0),
                                    layer.N, layer.relu, pool_group, _stream(xyz)), "sa_layer12")
    return out


def _premul_layers(first, Cf):
    """Derived packings of a set-abstraction block's first layer (columns [feature | xyz]) for the
    per-source-point evaluation: ``u`` = scale * W [f | x] (no shift, no ReLU) and ``v`` = scale * Wx x_c - shift."""
    if first.premul is None:
        u, v = _Layer(), _Layer()
        u.W, u.W8, u.premul, u.scale, u.N, u.K, u.Kpad, u.relu = first.W, None, None, first.scale, first.N, first.K, first.Kpad, 0
        u.shift = torch.zeros_like(first.shift)
        Wx = torch.zeros((first.W.shape[0], 16), dtype=torch.float32, device=first.W.device)
        Wx[:, :3] = first.W[:, Cf:Cf + 3]
        v.W, v.W8, v.premul, v.scale, v.N, v.K, v.Kpad, v.relu = Wx.contiguous(), None, None, first.scale, first.N, 3, 16, 0
        v.shift = (-first.shift).contiguous()
        first.premul = (u, v)
    return first.premul


@_on_tensor_device
def pack_rows(feature, xyz, width, mu=None):
    """Channels-last rows [feature | xyz - mu | 0] of every point: feature (B,Cf,N) or None, xyz (B,3,N), mu (B,3,1) or None
    -> (B*N, width)."""
    B, _, N = xyz.shape
    out = torch.empty((B * N, width), dtype=torch.float32, device=xyz.device)
    if feature is None:
        fptr, fb, fc, fn, Cf = None, 0, 0, 0, 0
    else:
        fptr, (fb, fc, fn), Cf = feature.data_ptr(), feature.stride(), feature.size(1)
    if mu is not None:
        mu = mu.reshape(B, 3)
        mu = mu if mu.is_contiguous() else mu.contiguous()
    _check(_L.regnet_pack_rows_centred_f32(fptr, fb, fc, fn, Cf, xyz.data_ptr(), *xyz.stride(),
                                           None if mu is None else mu.data_ptr(), B, N, width, out.data_ptr(),
                                           _stream(xyz)), "pack_rows")
    return out


@_on_tensor_device
def sa_premul_layer(U, V, nbr, layer, B, Nsrc, M, group, pool_group=0):
    """Layer 2 of a set-abstraction block over pre-multiplied layer-1 rows: relu(U[nbr] - V[centre]) . W."""
    P = B * M * group
    rows = P // pool_group if pool_group else P
    out = torch.empty((rows, layer.N), dtype=torch.float32, device=U.device)
    _check(_L.regnet_sa_premul_layer_f32(U.data_ptr(), U.stride(0), V.data_ptr(), V.stride(0), U.size(1),
                                         nbr.data_ptr(), B, Nsrc, M, group, layer.W.data_ptr(), layer.Kpad,
                                         layer.scale.data_ptr(), layer.shift.data_ptr(), out.data_ptr(),
                                         out.stride(0), layer.N, layer.relu, pool_group, _stream(U)), "sa_premul_layer")
    return out


def chain3_order(count):
    """Processing order of sa_chain3's neighbourhoods by cost class (<= 32 members, 33..48, more), stable within a class.
    (The expensive class first instead: 994-1 005 against 995-1 000 scenes/s over three alternations -- no difference.)"""
    c = count.reshape(-1)
    if c.is_cuda and c.dtype == torch.int64 and c.numel() <= (1 << 24):
        return pn2_ext.class_order(c)       # a three-bin counting sort in one launch (six tensor ops + a radix sort before)
    return torch.argsort((c > 32).to(torch.uint8) + (c > 48).to(torch.uint8), stable=True)


# EXPERIMENT, off: the level-1 block with fp32-faithful products on the bf16 matrix pipe (csrc/sa_split.hip).  Never a
# default path; bench.py reports it under its own name (value_split_products).
SPLIT_PRODUCTS = False
_split_planes = {}


@_on_tensor_device
def sa_chain3_split(feature, xyz, nbr, ctr, l1, l2, l3, B, M, group, count=None, order=None):
    """``sa_chain3`` by ``regnet_sa_chain3_split_f32``; the weights' bf16 pieces are built once per packed-weight tensor pair."""
    out = torch.empty((B * M, l3.N), dtype=torch.float32, device=xyz.device)
    if feature is None:
        fptr, fb, fn, fc, Cf = None, 0, 0, 0, 0
    else:
        fptr, (fb, fc, fn), Cf = feature.data_ptr(), feature.stride(), feature.size(1)
    key = (l2.W.data_ptr(), l2.W._version, l3.W.data_ptr(), l3.W._version)
    planes = _split_planes.get(key)
    build = planes is None
    if build:
        _split_planes.clear()
        planes = torch.empty((_L.regnet_sa_chain3_split_plane_bytes(l3.N),), dtype=torch.uint8, device=xyz.device)
        _split_planes[key] = planes
    _check(_L.regnet_sa_chain3_split_f32(fptr, fb, fn, fc, Cf, xyz.data_ptr(), *xyz.stride(), nbr.data_ptr(), ctr.data_ptr(),
                                         None if count is None else count.data_ptr(), None if order is None else order.data_ptr(), B, M,
                                         group, l1.W8.data_ptr(), l1.scale.data_ptr(), l1.shift.data_ptr(), l2.W.data_ptr(), l2.Kpad,
                                         l2.scale.data_ptr(), l2.shift.data_ptr(), l3.W.data_ptr(), l3.Kpad, l3.scale.data_ptr(),
                                         l3.shift.data_ptr(), l3.N, l3.relu, planes.data_ptr(), int(build), out.data_ptr(),
                                         out.stride(0), _tickets(xyz.device).data_ptr(), _stream(xyz)), "sa_chain3_split")
    return out


@_on_tensor_device
def sa_chain3(feature, xyz, nbr, ctr, l1, l2, l3, B, M, group, count=None, order=None):
    """Whole narrow-input SA block (gather, three layers, max over the neighbours) in one kernel; -> (B*M, C3).
    ``count`` (B,M) int64: members per neighbourhood (half the work for those with <= 32); ``order`` (B*M,) int64:
    processing order of the neighbourhoods (small ones together)."""
    if (SPLIT_PRODUCTS and group == 64 and l2.Kpad == 128 and l3.Kpad == 128 and l3.N == 256 and feature is not None
            and feature.size(1) == 3):      # (the shapes of the level-1 block, the only ones the experiment's kernel is built for)
        return sa_chain3_split(feature, xyz, nbr, ctr, l1, l2, l3, B, M, group, count, order)
    out = torch.empty((B * M, l3.N), dtype=torch.float32, device=xyz.device)
    if feature is None:
        fptr, fb, fn, fc, Cf = None, 0, 0, 0, 0
    else:
        fptr, (fb, fc, fn), Cf = feature.data_ptr(), feature.stride(), feature.size(1)
    _check(_L.regnet_sa_chain3_f32(fptr, fb, fn, fc, Cf, xyz.data_ptr(), *xyz.stride(), nbr.data_ptr(), ctr.data_ptr(),
                                   None if count is None else count.data_ptr(),
                                   None if order is None else order.data_ptr(), B, M, group, l1.W8.data_ptr(), l1.scale.data_ptr(), l1.shift.data_ptr(), l1.N,
                                   l2.W.data_ptr(), l2.Kpad, l2.scale.data_ptr(), l2.shift.data_ptr(), l2.N,
                                   l3.W.data_ptr(), l3.Kpad, l3.scale.data_ptr(), l3.shift.data_ptr(), l3.N, l3.relu,
                                   out.data_ptr(), out.stride(0), _stream(xyz)), "sa_chain3")
    return out


@_on_tensor_device
def interp_concat(sparse_cl, idx, dist2, eps, dense_feature, B, Nd):
    """sparse_cl: (B,Ns,Cs) channels-last contiguous; dense_feature (B,Cd,Nd) any strides or None.
    Returns the (B*Nd, round_up(Cs+Cd,4)) channels-last operand of the first FP layer and its valid width."""
    Cs = sparse_cl.size(2)
    Cd = 0 if dense_feature is None else dense_feature.size(1)
    width = _round_up(Cs + Cd, 4)
    out = torch.empty((B * Nd, width), dtype=torch.float32, device=sparse_cl.device)
    if dense_feature is None:
        dptr, db, dn, dc = None, 0, 0, 0
    else:
        dptr, (db, dc, dn) = dense_feature.data_ptr(), dense_feature.stride()
    _check(_L.regnet_interp_concat_f32(sparse_cl.data_ptr(), sparse_cl.stride(0), sparse_cl.stride(1), Cs,
                                       idx.data_ptr(), dist2.data_ptr(), float(eps), dptr, db, dn, dc, Cd, B, Nd,
                                       out.data_ptr(), out.stride(0), width, _stream(sparse_cl)), "interp_concat")
    return out, width


def _fp_split_layers(first, Cs):
    """Derived packings of a feature-propagation block's first layer (columns [interpolated | skip]):
    ``s`` = W[:, :Cs] and ``d`` = W[:, Cs:], both without affine / ReLU (applied after the interpolation)."""
    if first.premul is None:
        def part(cols):
            L = _Layer()
            K = cols.shape[1]
            L.N, L.K, L.Kpad, L.relu, L.W8, L.premul = first.N, K, _round_up(max(K, 1), 16), 0, None, None
            Wp = torch.zeros((first.W.shape[0], L.Kpad), dtype=torch.float32, device=first.W.device)
            Wp[:, :K] = cols
            L.W = Wp.contiguous()
            L.scale = torch.ones_like(first.scale)
            L.shift = torch.zeros_like(first.shift)
            return L
        Cd = first.K - Cs
        wd4 = None
        if 0 < Cd <= 4:   # narrow skip input (rgb): multiplied inside the interpolation kernel
            wd4 = torch.zeros((first.N, 4), dtype=torch.float32, device=first.W.device)
            wd4[:, :Cd] = first.W[:first.N, Cs:first.K]
            wd4 = wd4.contiguous()
        first.premul = (part(first.W[:, :Cs]), part(first.W[:, Cs:first.K]) if Cd > 4 else None, wd4)
    return first.premul[:3]


def _interp_tables(first, wd4):
    """Wd4 (256 x 4) | scale1 (256) | shift1 (256): the tables of fp_head_chain_interp's prologue, cached on the layer."""
    if len(first.premul) < 4:
        wd = wd4 if wd4 is not None else torch.zeros((256, 4), dtype=torch.float32, device=first.W.device)
        first.premul = tuple(first.premul[:3]) + (torch.cat([wd.reshape(-1), first.scale[:256], first.shift[:256]]).contiguous(),)
    return first.premul[3]


@_on_tensor_device
def interp_affine(Ys, idx, dist2, eps, Yd, dense_small, wd4, layer, B, Ns, Nd):
    """relu(scale * (sum_k w_k Ys[idx_k] + Yd + Wd4 . dense_small) + shift): the 3-NN interpolation of
    pre-multiplied sparse rows.  ``dense_small``: (B,Cd<=4,Nd) any strides, or None."""
    C = layer.N
    out = torch.empty((B * Nd, C), dtype=torch.float32, device=Ys.device)
    if dense_small is None:
        dptr, db, dc, dn, Cd, wptr = None, 0, 0, 0, 0, None
    else:
        dptr, (db, dc, dn), Cd, wptr = dense_small.data_ptr(), dense_small.stride(), dense_small.size(1), wd4.data_ptr()
    _check(_L.regnet_interp_affine_f32(Ys.data_ptr(), Ns * Ys.stride(0), Ys.stride(0), idx.data_ptr(),
                                       dist2.data_ptr(), float(eps), None if Yd is None else Yd.data_ptr(),
                                       0 if Yd is None else Yd.stride(0), dptr, db, dn, dc, Cd, wptr,
                                       layer.scale.data_ptr(), layer.shift.data_ptr(), layer.relu, B, Nd, C,
                                       out.data_ptr(), out.stride(0), _stream(Ys)), "interp_affine")
    return out


def _packed_head(seg):
    """conv_score weight + folded bn_score scalars, cached like the layer stacks (reading the
    scalars costs a device sync, so do it once per weight version, not per forward)."""
    sig = _signature(seg.conv_score) + _signature(seg.bn_score)
    cache = getattr(seg, "_regnet_head", None)
    if cache is None or cache[0] != sig:
        conv, bn = seg.conv_score, seg.bn_score
        w = conv.weight.detach().reshape(-1).float().contiguous()
        bn_scale = float(bn.weight.detach()[0] / torch.sqrt(bn.running_var.detach()[0] + bn.eps))
        bn_shift = float(bn.bias.detach()[0] - bn.running_mean.detach()[0] * bn_scale)
        bias = float(conv.bias.detach()[0]) if conv.bias is not None else 0.0
        cache = (sig, (w, bias, bn_scale, bn_shift))
        seg._regnet_head = cache
    return cache[1]


@_on_tensor_device
def score_head(x, seg, P):
    w, bias, bn_scale, bn_shift = _packed_head(seg)
    score = torch.empty((P,), dtype=torch.float32, device=x.device)
    _check(_L.regnet_score_head_f32(x.data_ptr(), x.stride(0), w.numel(), w.data_ptr(), bias, bn_scale, bn_shift,
                                    score.data_ptr(), P, _stream(x)), "score_head")
    return score


def _flop_meta(P, K, N):
    return "P%d K%d N%d flop%d" % (P, K, N, 2 * P * K * N)


# what bench.py brackets with HIP events: name -> meta(args) (algorithmic flops use the TRUE K)
TIMED_OPS = {
    "mlp_layer": lambda A, Ka, layer, P, pool_group=0: _flop_meta(P, layer.K, layer.N) + (" pool%d" % pool_group if pool_group else ""),
    "sa_layer1": lambda feature, xyz, nbr, ctr, layer, B, M, group: _flop_meta(B * M * group, layer.K, layer.N),
    "sa_premul_layer": lambda U, V, nbr, layer, B, Nsrc, M, group, pool_group=0:
        _flop_meta(B * M * group, layer.K, layer.N),
    "sa_chain3": lambda feature, xyz, nbr, ctr, l1, l2, l3, B, M, group, count=None, order=None:
        "P%d K%d N%d flop%d" % (B * M * group, l3.K, l3.N,
                                2 * B * M * group * (l1.K * l1.N + l2.K * l2.N + l3.K * l3.N)),
    "sa_premul_chain": lambda U, V, nbr, module, layers, B, Nsrc, M:
        "P%d K256 N512 flop%d" % (B * M * 64, 2 * B * M * 64 * (256 * 256 + 256 * 512)),
    "sa3_premul_chain": lambda U, V, nbr, module, layers, B, Nsrc, M:
        "P%d K512 N1024 flop%d" % (B * M * 64, 2 * B * M * 64 * (512 * 512 + 512 * 1024)),
    "fp_head_chain_interp": lambda Ys, idx, dist2, eps, dense_small, wd4, first, seg, fp_layers, B, Ns, Nd:
        "P%d K256 N256 flop%d" % (B * Nd, 2 * B * Nd * 491520),
    "fp_head_chain": lambda h1, seg, fp_layers, P: "P%d K256 N256 flop%d" % (P, 2 * P * 491520),
    "sa_layer12": lambda feature, xyz, nbr, ctr, first, layer, B, M, group, pool_group=0:
        "P%d K%d N%d flop%d" % (B * M * group, layer.K, layer.N,
                                2 * B * M * group * (layer.K * layer.N + first.K * first.N)),
}


def _as_channels_last(feature):
    """(B,C,N) tensor -> (B,N,C) contiguous view/copy with channel stride 1."""
    cl = feature.transpose(1, 2)
    return cl if cl.is_contiguous() else cl.contiguous()


# ---- block-level forwards ---------------------------------------------------------------------
# Every block is split into its GEOMETRY part (indices; depends on xyz only) and its FEATURE part
# (the MLP chain).  Called back to back they are the module forward; pipeline.ForwardPipeline runs
# the geometry of batch i+1 on another stream while the features of batch i are on the matrix cores.
def sa_sample(module, xyz, prefix_ok=None):
    """Furthest point sampling of a PointNetSAModule (modules.py:23-26): the long latency chain.  -> (ctr, first_tie).
    ``prefix_ok``: the ``first_tie`` of the sampling run that produced ``xyz``'s ORDER when ``xyz`` is the previous level's
    centroids in pick order (never anything else): scenes whose run had no tie among its first M picks then get 0 .. M-1
    without sampling -- the same indices the sampling would return (pn2_ext.FpsChain, csrc/geometry.hip)."""
    chain = pn2_ext.FpsChain(prefix_ok if FPS_CHAIN else None)
    ctr = pn2_ext.farthest_point_sample(xyz, module.num_centroids, chain)
    return ctr, chain.first_tie


def sa_group(module, xyz, ctr, first_tie=None):
    """Centroid gather + ball query given the sampled indices (modules.py:41, :238-239).  ``first_tie``: of the run that
    sampled ``ctr`` (kept in the plan: the next level's ``prefix_ok``)."""
    B, M = ctr.shape
    new_xyz = pn2_ext.gather_points(xyz, ctr) if xyz.dtype == torch.float32 else torch.gather(xyz, 2, ctr[:, None, :].expand(B, 3, M))
    nbr, count = pn2_ext.ball_query(xyz, new_xyz, module.grouper.radius, module.grouper.num_neighbours)
    geo = {"ctr": ctr, "new_xyz": new_xyz, "nbr": nbr}
    if first_tie is not None:
        geo["first_tie"] = first_tie
    if (CHAIN3 and module.grouper.num_neighbours == 64 and len(module.mlp) == 3
            and module.mlp[0].conv.in_channels <= 8):
        # for the register-chained block (narrow gathered input, sa_features): neighbourhoods with <= 32 members first
        # (one point tile instead of two), then those with 33..48 (waves w and w + 4 of a workgroup share a third tile:
        # three tiles per pair, csrc/sa_chain.hip), then the full ones, so that whole workgroups are of one kind;
        # computed here, in the geometry stage, off the matrix cores' critical path
        geo["count"] = count
        geo["order"] = chain3_order(count)
    elif (PREMUL and not module.training and not torch.is_grad_enabled() and module.use_xyz
          and module.mlp[0].conv.in_channels > 8 and len(module.mlp) >= 2):
        # wide gathered input (levels 2+, see sa_features): everything of the pre-multiplied first layer that depends on the
        # coordinates only -- the scene mean, the centred source coordinates, the per-centre term V -- is computed HERE, in
        # the geometry stage, off the feature stage's stream (two reductions / subtractions, a pack and a tiny GEMM per level)
        Cf = module.mlp[0].conv.in_channels - 3
        layers = _packed_stack(module, module.mlp,
                               lambda: torch.cat([torch.arange(3, 3 + Cf), torch.arange(0, 3)]).to(xyz.device))
        first = layers[0]
        if first.relu and first.N % 4 == 0 and layers[1].K == first.N:
            _, v_layer = _premul_layers(first, Cf)
            # (the subtraction of the scene mean happens inside the packs: pack_rows(..., mu))
            geo["mu"] = xyz.mean(dim=2, keepdim=True) if PREMUL_CENTRE else None
            geo["V"] = mlp_layer(pack_rows(None, new_xyz, 4, geo["mu"]), 4, v_layer, B * M)
            # V = W_xyz . centre with the first layer's BatchNorm folded in: a function of the WEIGHTS too, unlike the rest
            # of the plan.  sa_features recomputes it when the block's weights changed since (a plan reused across an
            # optimizer step / load_state_dict)
            geo["V_signature"] = _signature(module.mlp)
    return geo


def sa_geometry(module, xyz, ctr=None, prefix_ok=None):
    """FPS + centroid gather + ball query of a PointNetSAModule.  ``prefix_ok``: see sa_sample (``xyz`` must then be the
    previous level's ``new_xyz``)."""
    if ctr is not None:
        return sa_group(module, xyz, ctr)
    ctr, first_tie = sa_sample(module, xyz, prefix_ok)
    return sa_group(module, xyz, ctr, first_tie)


def sa_features(module, xyz, feature, geo):
    """Grouping + SharedMLP + max over K of a PointNetSAModule (modules.py:44-56, :244-245)."""
    B = xyz.shape[0]
    M, K = module.num_centroids, module.grouper.num_neighbours
    Cf = 0 if feature is None else feature.size(1)
    if K != 64 or len(module.mlp) < 2:
        raise NotImplementedError("fused SA supports 64 neighbours and >= 2 MLP layers")
    if feature is not None and not module.use_xyz:
        raise NotImplementedError("fused SA without use_xyz")
    # reference channel order is [xyz(3) | feature] (modules.py:52); the kernel gathers
    # [feature | xyz], so permute the first layer's weight columns accordingly.
    layers = _packed_stack(module, module.mlp,
                           lambda: torch.cat([torch.arange(3, 3 + Cf), torch.arange(0, 3)]).to(xyz.device))
    P = B * M * K
    if layers[0].W8 is not None and layers[0].N % 16 == 0 and layers[0].relu:
        # narrow gathered input (level 1: rgb + xyz): layer 1 is recomputed on the VALU inside layer 2's
        # operand load, its (P x C1) activation never exists in HBM
        if (CHAIN3 and len(layers) == 3 and layers[0].N == 128 and layers[1].N == 128 and layers[1].relu
                and layers[2].N % 32 == 0):
            # the whole block in one kernel, activations in registers (csrc/sa_chain.hip)
            pooled = sa_chain3(feature, xyz, geo["nbr"], geo["ctr"], layers[0], layers[1], layers[2], B, M, K,
                               geo.get("count"), geo.get("order"))
            return geo["new_xyz"], pooled.view(B, M, -1).transpose(1, 2)
        if len(layers) == 2:
            pooled = sa_layer12(feature, xyz, geo["nbr"], geo["ctr"], layers[0], layers[1], B, M, K, pool_group=K)
            return geo["new_xyz"], pooled.view(B, M, -1).transpose(1, 2)
        h = sa_layer12(feature, xyz, geo["nbr"], geo["ctr"], layers[0], layers[1], B, M, K)
        rest = layers[2:]
    elif PREMUL and feature is not None and layers[0].relu and layers[0].N % 4 == 0 and layers[1].K == layers[0].N:
        # wide gathered input (levels 2+): layer 1 is linear in [f_j | x_j - x_c], so it is evaluated once
        # per source point and once per centre (N + M rows instead of M * 64) and the gather, the
        # subtraction and the ReLU happen inside layer 2's operand load
        first, N1 = layers[0], xyz.shape[2]
        u_layer, v_layer = _premul_layers(first, Cf)
        width = _round_up(Cf + 3, 4)
        # U[j] - V[c] = s W [f_j | x_j - x_c] - t only up to fp32 rounding of the two big terms: both sides use the
        # coordinates RELATIVE TO THE SCENE'S MEAN (the difference is unchanged, the magnitudes -- table-top scenes sit
        # ~0.75 m from the origin -- and with them the cancellation error of the subtraction shrink)
        if "V" in geo and geo.get("V_signature") == _signature(module.mlp):
            mu, V = geo["mu"], geo["V"]                  # already done by the geometry stage (sa_group), same weights
        elif "V" in geo:                                 # stale: the weights moved after the plan was made
            mu = geo["mu"]
            V = mlp_layer(pack_rows(None, geo["new_xyz"], 4, mu), 4, v_layer, B * M)
        else:
            mu = xyz.mean(dim=2, keepdim=True) if PREMUL_CENTRE else None
            V = mlp_layer(pack_rows(None, geo["new_xyz"], 4, mu), 4, v_layer, B * M)
        U = mlp_layer(pack_rows(feature, xyz, width, mu), width, u_layer, B * N1)
        if supports_sa3_chain(layers) and U.size(1) == 512:
            # level 3: the same with 512-wide activations (layer 2 as two K-halves, csrc/rowchain.hip)
            pooled = sa3_premul_chain(U, V, geo["nbr"], module, layers, B, N1, M)
            return geo["new_xyz"], pooled.view(B, M, -1).transpose(1, 2)
        if supports_sa_chain(layers) and U.size(1) == 256:
            # layers 2 and 3 + the pooling in one kernel, layer-2 activation in registers (csrc/rowchain.hip)
            pooled = sa_premul_chain(U, V, geo["nbr"], module, layers, B, N1, M)
            return geo["new_xyz"], pooled.view(B, M, -1).transpose(1, 2)
        if len(layers) == 2:
            pooled = sa_premul_layer(U, V, geo["nbr"], layers[1], B, N1, M, K, pool_group=K)
            return geo["new_xyz"], pooled.view(B, M, -1).transpose(1, 2)
        h = sa_premul_layer(U, V, geo["nbr"], layers[1], B, N1, M, K)
        rest = layers[2:]
    else:
        h = sa_layer1(feature, xyz, geo["nbr"], geo["ctr"], layers[0], B, M, K)
        rest = layers[1:]
    for layer in rest[:-1]:
        h = mlp_layer(h, layer.K, layer, P)
    pooled = mlp_layer(h, rest[-1].K, rest[-1], P, pool_group=K)              # (B*M, C_out)
    return geo["new_xyz"], pooled.view(B, M, -1).transpose(1, 2)


def sa_forward(module, xyz, feature, geo=None):
    """PointNetSAModule.forward (modules.py:210-246) for num_centroids > 0 with a ball grouper."""
    return sa_features(module, xyz, feature, geo if geo is not None else sa_geometry(module, xyz))


def fp_geometry(module, dense_xyz, sparse_xyz):
    """3-NN search of a PointnetFPModule (modules.py:115-116)."""
    idx, dist2 = pn2_ext.point_search(dense_xyz, sparse_xyz, module.interpolator.num_neighbors)
    return {"idx": idx, "dist2": dist2}


def _fp_first_layer(module, layers, dense_xyz, dense_feature, sparse_feature, geo):
    """First layer of a feature-propagation block evaluated on the SPARSE rows (see fp_features); -> (B*Nd, C1)
    channels-last activation after BN + ReLU, or None when the split does not apply."""
    B, _, Nd = dense_xyz.shape
    Cs, Ns = sparse_feature.size(1), sparse_feature.size(2)
    Cd = 0 if dense_feature is None else dense_feature.size(1)
    first = layers[0]
    if not (PREMUL and first.N % 4 == 0 and 256 % (first.N // 4) == 0 and Cs % 4 == 0 and (Cd % 4 == 0 or Cd <= 4)
            and Ns < Nd and first.K == Cs + Cd and (dense_feature is None or dense_feature.dtype == torch.float32)):
        return None
    # layer 1 is linear in [interpolated | skip]: multiply the SPARSE rows (and the skip rows by their own
    # weight columns), interpolate the products
    lay_s, lay_d, wd4 = _fp_split_layers(first, Cs)
    Ys = mlp_layer(_as_channels_last(sparse_feature).view(B * Ns, Cs), Cs, lay_s, B * Ns)
    Yd = None
    if lay_d is not None:
        Yd = mlp_layer(_as_channels_last(dense_feature).view(B * Nd, Cd), Cd, lay_d, B * Nd)
    return interp_affine(Ys, geo["idx"], geo["dist2"], module.interpolator._eps, Yd,
                         dense_feature if wd4 is not None else None, wd4, first, B, Ns, Nd)


def fp_features(module, dense_xyz, dense_feature, sparse_feature, geo):
    """Interpolate + concat + SharedMLP of a PointnetFPModule (modules.py:117-131, :507)."""
    B, _, Nd = dense_xyz.shape
    layers = _packed_stack(module, module.mlp)
    h = _fp_first_layer(module, layers, dense_xyz, dense_feature, sparse_feature, geo)
    if h is not None:
        Ka, P = layers[0].N, B * Nd
        for layer in layers[1:]:
            h = mlp_layer(h, Ka, layer, P)
            Ka = layer.N
        return h.view(B, Nd, -1).transpose(1, 2)
    A, width = interp_concat(_as_channels_last(sparse_feature), geo["idx"], geo["dist2"], module.interpolator._eps,
                             dense_feature, B, Nd)
    P = B * Nd
    h, Ka = A, width
    for layer in layers:
        h = mlp_layer(h, Ka, layer, P)
        Ka = layer.N
    return h.view(B, Nd, -1).transpose(1, 2)


def fp_forward(module, dense_xyz, sparse_xyz, dense_feature, sparse_feature, geo=None):
    """PointnetFPModule.forward with a 3-NN interpolator (modules.py:104-131, :500-509)."""
    if geo is None:
        geo = fp_geometry(module, dense_xyz, sparse_xyz)
    return fp_features(module, dense_xyz, dense_feature, sparse_feature, geo)


def head_forward(seg, sparse_feature):
    """PointNet2Seg head (pointnet2.py:116-119): SharedMLP -> conv_score -> bn_score -> sigmoid -> (B,N)."""
    B, C, N = sparse_feature.shape
    layers = _packed_stack(seg.mlp, seg.mlp)
    h = _as_channels_last(sparse_feature).view(B * N, C)
    Ka = C
    if Ka % 4:
        raise NotImplementedError("head input width must be a multiple of 4")
    for layer in layers:
        h = mlp_layer(h, Ka, layer, B * N)
        Ka = layer.N
    return score_head(h, seg, B * N).view(B, N)


# ---- level-2 set-abstraction block: layers 2 + 3 + pooling as ONE kernel (csrc/rowchain.hip) ---------------------------
def _packed_sa_chain(module, layers):
    """Weight stream (24 stages [32 rows][256 k], layer 2 then layer 3) + affine table of sa_premul_chain."""
    sig = _signature(module.mlp)
    cache = getattr(module, "_regnet_sa_chain", None)
    if cache is None or cache[0] != sig:
        l2, l3 = layers[1], layers[2]
        stages = [_swizzle_stage(l2.W[32 * s:32 * s + 32, :256]) for s in range(8)]
        stages += [_swizzle_stage(l3.W[32 * s:32 * s + 32, :256]) for s in range(16)]
        stream = torch.cat(stages).contiguous()
        affine = torch.cat([l2.scale[:256], l2.shift[:256], l3.scale[:512], l3.shift[:512]]).contiguous()
        assert stream.numel() == _L.regnet_sa_premul_chain_stream_floats() and affine.numel() == 1536
        cache = (sig, (stream, affine))
        module._regnet_sa_chain = cache
    return cache[1]


def supports_sa_chain(layers):
    return (ROWCHAIN and len(layers) == 3 and layers[0].N == 256 and layers[1].K == 256 and layers[1].N == 256
            and layers[1].relu and layers[2].K == 256 and layers[2].N == 512)


@_on_tensor_device
def sa_premul_chain(U, V, nbr, module, layers, B, Nsrc, M):
    """relu(U[nbr] - V[centre]) -> 256 -> 512 -> max over the 64 neighbours, one launch; -> (B*M, 512)."""
    stream, affine = _packed_sa_chain(module, layers)
    out = torch.empty((B * M, 512), dtype=torch.float32, device=U.device)
    ticket = _tickets(U.device)
    _check(_L.regnet_sa_premul_chain_f32(U.data_ptr(), U.stride(0), V.data_ptr(), V.stride(0), nbr.data_ptr(), B, Nsrc,
                                         M, stream.data_ptr(), 24, affine.data_ptr(), affine.numel(), layers[2].relu,
                                         out.data_ptr(), out.stride(0), ticket.data_ptr(), _stream(U)),
           "sa_premul_chain")
    return out


# ---- level-3 set-abstraction block: layers 2 + 3 + pooling as ONE kernel, 512-wide (csrc/rowchain.hip) ------------------
def _packed_sa3_chain(module, layers):
    """Weight stream (96 stages [32 rows][256 k]: W2 as (K-half, 16 row blocks), W3 as (32 row blocks, K-half)) + affine
    table of sa3_premul_chain."""
    sig = _signature(module.mlp)
    cache = getattr(module, "_regnet_sa3_chain", None)
    if cache is None or cache[0] != sig:
        l2, l3 = layers[1], layers[2]
        stages = [_swizzle_stage(l2.W[32 * rg:32 * rg + 32, 256 * kh:256 * kh + 256]) for kh in range(2) for rg in range(16)]
        stages += [_swizzle_stage(l3.W[32 * s:32 * s + 32, 256 * kh:256 * kh + 256]) for s in range(32) for kh in range(2)]
        stream = torch.cat(stages).contiguous()
        affine = torch.cat([l2.scale[:512], l2.shift[:512], l3.scale[:1024], l3.shift[:1024]]).contiguous()
        assert stream.numel() == _L.regnet_sa3_premul_chain_stream_floats() and affine.numel() == 3072
        cache = (sig, (stream, affine))
        module._regnet_sa3_chain = cache
    return cache[1]


def supports_sa3_chain(layers):
    return (SA3_CHAIN and len(layers) == 3 and layers[0].N == 512 and layers[1].K == 512 and layers[1].N == 512
            and layers[1].relu and layers[2].K == 512 and layers[2].N == 1024)


@_on_tensor_device
def sa3_premul_chain(U, V, nbr, module, layers, B, Nsrc, M):
    """relu(U[nbr] - V[centre]) -> 512 -> 1024 -> max over the 64 neighbours, one launch; -> (B*M, 1024)."""
    stream, affine = _packed_sa3_chain(module, layers)
    out = torch.empty((B * M, 1024), dtype=torch.float32, device=U.device)
    ticket = _tickets(U.device)
    _check(_L.regnet_sa3_premul_chain_f32(U.data_ptr(), U.stride(0), V.data_ptr(), V.stride(0), nbr.data_ptr(), B, Nsrc,
                                          M, stream.data_ptr(), 96, affine.data_ptr(), affine.numel(), layers[2].relu,
                                          out.data_ptr(), out.stride(0), ticket.data_ptr(), _stream(U)),
           "sa3_premul_chain")
    return out


# ---- FP3 tail + segmentation head as ONE kernel (csrc/rowchain.hip) --------------------------------------------------
ROWCHAIN = True   # last FP block's layers 2-3 + the whole head in one register-chained kernel
SA3_CHAIN = True  # level-3 set-abstraction block's layers 2-3 + pooling in one register-chained kernel


def _swizzle_stage(block):
    """(R, KC) weight block -> the LDS image of a rowchain stage: 16-byte chunk c' of row r holds logical chunk
    c' ^ (r & 15) (XOR on the low 4 bits of the chunk index; see csrc/rowchain.hip:rc_frag_offsets)."""
    R, KC = block.shape
    chunks = block.reshape(R, KC // 4, 4)
    r = torch.arange(R, device=block.device)[:, None]
    cp = torch.arange(KC // 4, device=block.device)[None, :]
    src = (cp & ~15) | ((cp ^ r) & 15)
    return chunks.gather(1, src[:, :, None].expand(R, KC // 4, 4)).reshape(-1)


def _rowchain_pair_stream(la, lb):
    """Stream of one layer pair (A: 256 -> M, B: M -> N) in consumption order: per 128-channel group of M, four
    A-stages [32 rows][256 k] then N/64 B-stages [64 rows][128 k]."""
    WA, WB = la.W[:la.N, :la.K], lb.W[:lb.N, :lb.K]
    M, N = la.N, lb.N
    assert la.K == 256 and lb.K == M and M % 128 == 0 and N % 64 == 0
    out = []
    for ob in range(M // 128):
        for u in range(4):
            out.append(_swizzle_stage(WA[128 * ob + 32 * u:128 * ob + 32 * u + 32, :]))
        for v in range(N // 64):
            out.append(_swizzle_stage(WB[64 * v:64 * v + 64, 128 * ob:128 * ob + 128]))
    return out


def _packed_rowchain(seg, fp_layers):
    """Weight stream + affine table of the FP3-tail + head chain, cached on ``seg`` per weight version."""
    head = _packed_stack(seg.mlp, seg.mlp)
    layers = list(fp_layers[1:]) + list(head)
    sig = _signature(seg.fp_modules[-1]) + _signature(seg.mlp)
    cache = getattr(seg, "_regnet_rowchain", None)
    if cache is None or cache[0] != sig:
        stages = []
        for a, b in ((layers[0], layers[1]), (layers[2], layers[3]), (layers[4], layers[5])):
            stages += _rowchain_pair_stream(a, b)
        stream = torch.cat(stages).contiguous()
        affine = torch.cat([torch.cat([L.scale[:L.N], L.shift[:L.N]]) for L in layers]).contiguous()
        assert stream.numel() == _L.regnet_fp_head_chain_stream_floats() and affine.numel() == 3328
        cache = (sig, (stream, affine))
        seg._regnet_rowchain = cache
    return cache[1]


def supports_rowchain(seg, fp_module):
    """The chained kernel covers exactly the reference's configuration (pointnet2.py:44-46): FP channels
    (256, 256, 256) and head channels (512, 256, 256, 128) with ReLU everywhere, one score channel."""
    if not ROWCHAIN or seg.k_score != 1:
        return False
    widths = [b.conv.out_channels for b in fp_module.mlp] + [b.conv.out_channels for b in seg.mlp]
    relus = [b.relu is not None and b.bn is not None for b in list(fp_module.mlp) + list(seg.mlp)]
    return widths == [256, 256, 256, 512, 256, 256, 128] and all(relus) and seg.mlp[0].conv.in_channels == 256


@_on_tensor_device
def fp_head_chain(h1, seg, fp_layers, P):
    """h1 (P, 256) -> (F (P, 256), score (P,)): FP layers 2-3 and the segmentation head in one launch."""
    stream, affine = _packed_rowchain(seg, fp_layers)
    w, bias, bn_scale, bn_shift = _packed_head(seg)
    F = torch.empty((P, 256), dtype=torch.float32, device=h1.device)
    score = torch.empty((P,), dtype=torch.float32, device=h1.device)
    ticket = _tickets(h1.device)   # work-queue head, zeroed on this stream
    _check(_L.regnet_fp_head_chain_f32(h1.data_ptr(), h1.stride(0), stream.data_ptr(), 60, affine.data_ptr(),
                                       affine.numel(), w.data_ptr(), bias, bn_scale, bn_shift, F.data_ptr(),
                                       F.stride(0), score.data_ptr(), P, ticket.data_ptr(), 0, -1, _stream(h1)),
           "fp_head_chain")
    return F, score


FP_HEAD_INTERP = True   # ... with the block's first layer (interpolation of the pre-multiplied sparse rows) in its prologue
TAIL_SINK = None        # a list (set by ForwardPipeline around a feature stage): fp_head_chain_interp puts its partial last round
                        # of blocks on a side stream and appends the event that marks its end; None: one launch, as always
_CUS = 256
_tail_streams = {}


def _tail_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _tail_streams:
        _tail_streams[key] = torch.cuda.Stream(device)
    return _tail_streams[key]


@_on_tensor_device
def fp_head_chain_interp(Ys, idx, dist2, eps, dense_small, wd4, first, seg, fp_layers, B, Ns, Nd):
    """(Ys (B*Ns, 256) pre-multiplied sparse rows, 3-NN indices / squared distances, rgb-like skip input) -> (F, score):
    the whole last feature-propagation block + the segmentation head in one launch; h1 is never written."""
    stream, affine = _packed_rowchain(seg, fp_layers)
    w, bias, bn_scale, bn_shift = _packed_head(seg)
    tables = _interp_tables(first, wd4)
    P = B * Nd
    F = torch.empty((P, 256), dtype=torch.float32, device=Ys.device)
    score = torch.empty((P,), dtype=torch.float32, device=Ys.device)
    tickets = _tickets(Ys.device, 2)
    if dense_small is None:
        dptr, db, dc, dn, Cd = None, 0, 0, 0, 0
    else:
        dptr, (db, dc, dn), Cd = dense_small.data_ptr(), dense_small.stride(), dense_small.size(1)

    def launch(first, count, ticket, stream_handle):
        _check(_L.regnet_fp_head_chain_interp_f32(Ys.data_ptr(), Ns * Ys.stride(0), Ys.stride(0), idx.data_ptr(),
                                                  dist2.data_ptr(), float(eps), dptr, db, dn, dc, Cd, tables.data_ptr(), B,
                                                  Nd, stream.data_ptr(), 60, affine.data_ptr(), affine.numel(), w.data_ptr(),
                                                  bias, bn_scale, bn_shift, F.data_ptr(), F.stride(0), score.data_ptr(),
                                                  ticket.data_ptr(), first, count, stream_handle), "fp_head_chain_interp")

    # The last round of 128-row blocks is partial (8 x 25 600 rows: 1600 blocks = 6 x 256 + 64): run alone it keeps 64 CUs
    # busy for a whole pass while 192 idle (a pass costs the same however few blocks it holds).  A caller that can run
    # something else meanwhile -- ForwardPipeline: the NEXT batch's first kernels -- collects it from TAIL_SINK: the tail
    # goes to a side stream behind the inputs, and the caller orders the consumers of F / score behind the event it gets.
    blocks = _L.regnet_fp_head_chain_blocks(P)
    tail = blocks % _CUS
    if TAIL_SINK is not None and blocks > _CUS and 0 < tail <= _CUS // 2:
        cur = torch.cuda.current_stream(Ys.device)
        side = _tail_stream(Ys.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        side.wait_event(ready)
        launch(blocks - tail, tail, tickets[1:], side.cuda_stream)
        done = torch.cuda.Event()
        done.record(side)
        for t in (Ys, idx, dist2, tables, stream, affine, w, F, score, tickets) + (() if dense_small is None else (dense_small,)):
            t.record_stream(side)
        TAIL_SINK.append(done)
        launch(0, blocks - tail, tickets[:1], cur.cuda_stream)
    else:
        launch(0, -1, tickets[:1], _stream(Ys))
    return F, score


def fp_head_forward(seg, fp_module, dense_xyz, dense_feature, sparse_feature, geo):
    """Last feature-propagation block + segmentation head (pointnet2.py:64-84, :116-119) -> (feature (B,256,N) view,
    score (B,N)); None when the chained kernel does not apply (the caller then runs the two blocks separately)."""
    if not supports_rowchain(seg, fp_module):
        return None
    B, _, Nd = dense_xyz.shape
    layers = _packed_stack(fp_module, fp_module.mlp)
    Cs, Ns = sparse_feature.size(1), sparse_feature.size(2)
    Cd = 0 if dense_feature is None else dense_feature.size(1)
    first = layers[0]
    if (FP_HEAD_INTERP and PREMUL and first.N == 256 and first.relu and Cs % 4 == 0 and Cd <= 4 and Ns < Nd
            and first.K == Cs + Cd and sparse_feature.dtype == torch.float32
            and (dense_feature is None or dense_feature.dtype == torch.float32)):
        # the first layer on the SPARSE rows (fused._fp_first_layer), its interpolation inside the chain's prologue
        lay_s, _, wd4 = _fp_split_layers(first, Cs)
        Ys = mlp_layer(_as_channels_last(sparse_feature).view(B * Ns, Cs), Cs, lay_s, B * Ns)
        F, score = fp_head_chain_interp(Ys, geo["idx"], geo["dist2"], fp_module.interpolator._eps,
                                        dense_feature if wd4 is not None else None, wd4, first, seg, layers, B, Ns, Nd)
        return F.view(B, Nd, 256).transpose(1, 2), score.view(B, Nd)
    h1 = _fp_first_layer(fp_module, layers, dense_xyz, dense_feature, sparse_feature, geo)
    if h1 is None:
        return None
    F, score = fp_head_chain(h1, seg, layers, B * Nd)
    return F.view(B, Nd, 256).transpose(1, 2), score.view(B, Nd)


def prepack(score_net, region_net=None):
    """Build every packed-weight cache of the fused forward NOW, on the current stream.  The caches are otherwise
    filled lazily by whichever stream first runs a block; with several feature-stage streams
    (``ForwardPipeline(mlp_streams=2)``) a second stream could then read a cache whose packing kernels, enqueued on
    the first stream, have not finished.  ``ForwardPipeline.run`` calls this before its streams start."""
    forget(score_net, region_net)
    seg = getattr(score_net, "extrat_featurePN2", score_net)
    dev = next(seg.parameters()).device
    for sa in seg.sa_modules:
        Cf = sa.in_channels
        layers = _packed_stack(sa, sa.mlp, lambda Cf=Cf: torch.cat([torch.arange(3, 3 + Cf), torch.arange(0, 3)]).to(dev))
        if layers[0].W8 is None and Cf > 0:
            _premul_layers(layers[0], Cf)
            if supports_sa_chain(layers):
                _packed_sa_chain(sa, layers)
            if supports_sa3_chain(layers):
                _packed_sa3_chain(sa, layers)
    sparse = seg.sa_modules[-1].out_channels
    for fp in seg.fp_modules:
        layers = _packed_stack(fp, fp.mlp)
        Cs = sparse
        if Cs <= layers[0].K:
            split = _fp_split_layers(layers[0], Cs)
            if fp is seg.fp_modules[-1] and layers[0].N == 256:
                _interp_tables(layers[0], split[2])
        sparse = fp.out_channels
    _packed_stack(seg.mlp, seg.mlp)
    _packed_head(seg)
    if supports_rowchain(seg, seg.fp_modules[-1]):
        _packed_rowchain(seg, _packed_stack(seg.fp_modules[-1], seg.fp_modules[-1].mlp))
    if region_net is not None:
        _packed_named(region_net.extrat_feature_region, _TWOSTAGE)
        _packed_named(region_net.extrat_feature_refine, _REFINE)


def plan_tensors(plan):
    """All tensors of a geometry plan (for Tensor.record_stream when it crosses streams)."""
    out = []
    for level in plan["sa"] + plan["fp"]:
        out.extend(v for v in level.values() if isinstance(v, torch.Tensor))
    return out


# ---- grasp-region / refine heads (tiny GEMMs: P = #centres; kept on the same kernel so the whole
#      forward is deterministic and independent of the torch convolution backend) -----------------
def _packed_named(net, names):
    """names: list of (conv_attr, bn_attr, relu).  Cached on ``net`` like the SharedMLP stacks."""
    sig = _signature(net)
    cache = getattr(net, "_regnet_heads", None)
    if cache is None or cache[0] != sig:
        cache = (sig, {c: _pack(getattr(net, c), getattr(net, b), relu) for c, b, relu in names})
        net._regnet_heads = cache
    return cache[1]


_TWOSTAGE = [("conv", "bn", True), ("conv_cls2", "bn_cls2", True), ("conv_cls3", "bn_cls3", True),
             ("conv_cls4", "bn_cls4", False), ("conv_reg2", "bn_reg2", True), ("conv_reg3", "bn_reg3", True),
             ("conv_reg4", "bn_reg4", False)]
_REFINE = [("conv_formal", "bn_formal", True), ("conv_formal_cls2", "bn_formal_cls2", True),
           ("conv_formal_cls3", "bn_formal_cls3", False), ("conv_formal_reg2", "bn_formal_reg2", True),
           ("conv_formal_reg3", "bn_formal_reg3", False)]


FPS_CHAIN = True     # levels 2+ hand the level above's sampling certificate down (sa_sample): no sampling unless it had a tie
HEADS_CHAIN = True   # the grasp heads as ONE launch each (csrc/heads.hip) instead of a split-K GEMM + its reduction per layer
# ... up to this many rows.  The one launch is 16 rows per workgroup, each streaming all the head's weights through one CU for
# ~0.1-0.4 ms: the latency win at B <= 4 (<= 256 centres).  At B = 8 its 32 whole-CU workgroups sit under the NEXT batch's SA
# chains and the layer-wise split-K path measured 1 % faster end to end (scripts/ablate/frac_ab.sh, profiles/r04j_heads_ab.txt).
HEADS_CHAIN_MAX_ROWS = 256


def _heads_chain(x, L, plan, n_a, n_b):
    """x (n, K) float32 rows; ``plan``: [(layer name, src buffer, dst buffer)] in execution order (buffers: 0 input, 1-3 LDS
    scratch, 4 / 5 outputs a / b) -> (out_a (n, n_a), out_b (n, n_b)).  The int64 descriptor (device addresses of the packed
    weights) is built per call: nine integers per layer."""
    n = x.shape[0]
    x = x if x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 else x.contiguous().clone()
    descr = []
    for name, src, dst in plan:
        lay = L[name]
        descr += [lay.W.data_ptr(), lay.scale.data_ptr(), lay.shift.data_ptr(), lay.K, lay.Kpad, lay.N, lay.relu, src, dst]
    import ctypes
    arr = (ctypes.c_int64 * len(descr))(*descr)
    out_a = torch.empty((n, n_a), dtype=torch.float32, device=x.device)
    out_b = torch.empty((n, n_b), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(_L.regnet_heads_chain_f32(x.data_ptr(), x.stride(0), x.shape[1], n, ctypes.addressof(arr), len(plan),
                                         out_a.data_ptr(), n_a, out_b.data_ptr(), n_b, _stream(x)), "heads_chain")
    return out_a, out_b


HEADS_TREE = True    # more rows than HEADS_CHAIN_MAX_ROWS: ONE launch of csrc/heads.hip's heads_tree_kernel (32 rows per workgroup,
                     # the trunk activation chunked through LDS) instead of a split-K GEMM + its reduction per layer


def _heads_tree(net, x, L, trunk, first, tails, n_a, n_b):
    """x (n, K) rows -> (out_a (n, n_a), out_b (n, n_b)) by ``regnet_heads_tree_f32``.  ``trunk``: layer name; ``first``: the
    two branches' first layers (their packed weights / folded BatchNorm concatenated row-wise, cached on ``net`` with the
    packed layers' signature); ``tails``: [(layer, src, src_off, dst, dst_off)] in execution order."""
    n = x.shape[0]
    x = x if x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 else x.contiguous().clone()
    la, lb = L[first[0]], L[first[1]]
    cache = getattr(net, "_regnet_heads_tree", None)
    key = (la.W.data_ptr(), lb.W.data_ptr(), la.scale.data_ptr(), lb.scale.data_ptr())
    if cache is None or cache[0] != key:
        joined = (torch.cat([la.W[:la.N], lb.W[:lb.N]]).contiguous(), torch.cat([la.scale, lb.scale]).contiguous(),
                  torch.cat([la.shift, lb.shift]).contiguous())
        cache = (key, joined)
        net._regnet_heads_tree = cache
    W2, s2, t2 = cache[1]
    lt = L[trunk]
    descr = [lt.W.data_ptr(), lt.scale.data_ptr(), lt.shift.data_ptr(), lt.K, lt.Kpad, lt.N, lt.relu, 0, 0, 0, 0,
             W2.data_ptr(), s2.data_ptr(), t2.data_ptr(), lt.N, lt.N, la.N + lb.N, la.relu, 0, 0, 0, 0]
    for name, src, src_off, dst, dst_off in tails:
        lay = L[name]
        descr += [lay.W.data_ptr(), lay.scale.data_ptr(), lay.shift.data_ptr(), lay.K, lay.Kpad, lay.N, lay.relu, src, src_off,
                  dst, dst_off]
    import ctypes
    arr = (ctypes.c_int64 * len(descr))(*descr)
    out_a = torch.empty((n, n_a), dtype=torch.float32, device=x.device)
    out_b = torch.empty((n, n_b), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(_L.regnet_heads_tree_f32(x.data_ptr(), x.stride(0), x.shape[1], n, ctypes.addressof(arr), len(tails),
                                        out_a.data_ptr(), n_a, out_b.data_ptr(), n_b, _stream(x)), "heads_tree")
    return out_a, out_b


def _tree_ok(L, trunk, first, x):
    la, lb, lt = L[first[0]], L[first[1]], L[trunk]
    return (HEADS_TREE and x.shape[1] % 4 == 0 and lt.N % 256 == 0 and la.K == lb.K == lt.N and la.relu == lb.relu
            and la.N % 16 == 0 and la.N + lb.N in (256, 512))


_TWOSTAGE_PLAN = [("conv", 0, 1), ("conv_cls2", 1, 2), ("conv_cls3", 2, 3), ("conv_cls4", 3, 4),
                  ("conv_reg2", 1, 2), ("conv_reg3", 2, 3), ("conv_reg4", 3, 5)]
_REFINE_PLAN = [("conv_formal", 0, 1), ("conv_formal_cls2", 1, 2), ("conv_formal_cls3", 2, 4),
                ("conv_formal_reg2", 1, 2), ("conv_formal_reg3", 2, 5)]


def _chain(x, layers, names):
    P = x.shape[0]
    for n in names:
        x = mlp_layer(x, layers[n].K, layers[n], P)
    return x


def twostage_forward(net, mp_x, raw_reg=False):
    """PointNet2TwoStage.forward after the max-pool (pointnet2.py:174-188): mp_x (n, 256, 1) ->
    x_cls (n, k_cls), x_reg (n, k_cls, k_reg/k_cls) with sigmoid on channels 7: (``raw_reg``: without it -- the caller's
    decode kernel applies it to the one anchor it keeps, region_ops.stage2_decode)."""
    n = mp_x.shape[0]
    L = _packed_named(net, _TWOSTAGE)
    x = mp_x.reshape(n, -1).contiguous()
    if HEADS_CHAIN and n <= HEADS_CHAIN_MAX_ROWS and x.shape[1] % 16 == 0:
        x_cls, x_reg = _heads_chain(x, L, _TWOSTAGE_PLAN, L["conv_cls4"].N, L["conv_reg4"].N)
        x_reg = x_reg.view(n, -1, net.k_reg_no_anchor)
        if not raw_reg:
            x_reg[:, :, 7:] = torch.sigmoid(x_reg[:, :, 7:])
        return x_cls, x_reg
    if _tree_ok(L, "conv", ("conv_cls2", "conv_reg2"), x) and L["conv_cls3"].K == L["conv_cls2"].N:
        n2, n3 = L["conv_cls2"].N, L["conv_cls3"].N
        x_cls, x_reg = _heads_tree(net, x, L, "conv", ("conv_cls2", "conv_reg2"),
                                   [("conv_cls3", 0, 0, 1, 0), ("conv_reg3", 0, n2, 1, _round_up(n3, 16)),
                                    ("conv_cls4", 1, 0, 4, 0), ("conv_reg4", 1, _round_up(n3, 16), 5, 0)],
                                   L["conv_cls4"].N, L["conv_reg4"].N)
        x_reg = x_reg.view(n, -1, net.k_reg_no_anchor)
        if not raw_reg:
            x_reg[:, :, 7:] = torch.sigmoid(x_reg[:, :, 7:])
        return x_cls, x_reg
    h = mlp_layer(x, L["conv"].K, L["conv"], n)
    x_cls = _chain(h, L, ["conv_cls2", "conv_cls3", "conv_cls4"])
    x_reg = _chain(h, L, ["conv_reg2", "conv_reg3", "conv_reg4"]).view(n, -1, net.k_reg_no_anchor)
    if not raw_reg:
        x_reg[:, :, 7:] = torch.sigmoid(x_reg[:, :, 7:])
    return x_cls, x_reg


def refine_forward(net, x):
    """PointNet2Refine.forward after pooling + concat (pointnet2.py:240-253): x (n, 384, 1) ->
    x_cls (n, 2), x_reg (n, k_reg)."""
    n = x.shape[0]
    L = _packed_named(net, _REFINE)
    if HEADS_CHAIN and n <= HEADS_CHAIN_MAX_ROWS and (x.numel() // max(n, 1)) % 16 == 0:
        return _heads_chain(x.reshape(n, -1), L, _REFINE_PLAN, L["conv_formal_cls3"].N, L["conv_formal_reg3"].N)
    rows = x.reshape(n, -1)
    if _tree_ok(L, "conv_formal", ("conv_formal_cls2", "conv_formal_reg2"), rows):
        n2 = L["conv_formal_cls2"].N
        return _heads_tree(net, rows, L, "conv_formal", ("conv_formal_cls2", "conv_formal_reg2"),
                           [("conv_formal_cls3", 0, 0, 4, 0), ("conv_formal_reg3", 0, n2, 5, 0)],
                           L["conv_formal_cls3"].N, L["conv_formal_reg3"].N)
    h = mlp_layer(x.reshape(n, -1).contiguous(), L["conv_formal"].K, L["conv_formal"], n)
    return (_chain(h, L, ["conv_formal_cls2", "conv_formal_cls3"]),
            _chain(h, L, ["conv_formal_reg2", "conv_formal_reg3"]))
