"""The grasp heads of the region stage in TRAINING mode as one autograd node each (csrc/heads_train.hip).

Reference: PointNet2TwoStage.forward after its max-pool (multi_model/utils/pointnet2.py:174-188: conv 256 -> 1024, then a class
branch 1024 -> 256 -> 128 -> k_cls and a regression branch 1024 -> 256 -> 128 -> k_reg) and PointNet2Refine.forward
(pointnet2.py:240-253: conv 384 -> 1024, then 1024 -> 128 -> k_cls and 1024 -> 128 -> k_reg); every layer is
nn.Conv1d(.., 1) with bias + nn.BatchNorm1d on batch statistics, ReLU except on the last layer of a branch.

Through torch a layer is six launches forward and seven backward, and the stream of the region stage is paced by the host:
the twelve layers were ~160 of the ~400 launches between the end of ScoreNet's forward and the start of its backward, a
stretch in which the GPU has nothing else to run.  Here a head is ONE ``autograd.Function``: its forward calls
``regnet_head_layer_train_fwd_f32`` once per layer (batch statistics, running statistics, counter, normalisation and ReLU in
the launch that multiplies), its backward ``regnet_head_layer_train_bwd_f32`` once per layer (BatchNorm backward, bias /
weight gradients, then the input gradient, accumulated where two branches meet).  Intermediate activations live in one
workspace tensor; only the two leaf outputs are tensors of their own (the caller modifies the regression output in place).
"""
import torch

ENABLED = True     # module switch (bench.py --set heads_train.ENABLED=0 / tests): 0 -> layer by layer through torch
CALLS = {"forward": 0, "backward": 0}


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _layers_twostage(m):
    """(conv, bn, parent, relu) in execution order; parent -1 = the head's input."""
    return [(m.conv, m.bn, -1, True),
            (m.conv_cls2, m.bn_cls2, 0, True), (m.conv_cls3, m.bn_cls3, 1, True), (m.conv_cls4, m.bn_cls4, 2, False),
            (m.conv_reg2, m.bn_reg2, 0, True), (m.conv_reg3, m.bn_reg3, 4, True), (m.conv_reg4, m.bn_reg4, 5, False)]


def _layers_refine(m):
    return [(m.conv_formal, m.bn_formal, -1, True),
            (m.conv_formal_cls2, m.bn_formal_cls2, 0, True), (m.conv_formal_cls3, m.bn_formal_cls3, 1, False),
            (m.conv_formal_reg2, m.bn_formal_reg2, 0, True), (m.conv_formal_reg3, m.bn_formal_reg3, 3, False)]


def supported(layers, x):
    """x: (R, K, 1) or (R, K) float32 CUDA rows; every layer within regnet_head_layer_train_supported and plain BatchNorm1d
    (affine, running statistics, a momentum)."""
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() in (2, 3) and (x.dim() == 2 or x.shape[2] == 1)):
        return False
    from . import _lib
    R = x.shape[0]
    for conv, bn, _, _ in layers:
        N, K = conv.weight.shape[0], conv.weight.shape[1]
        if not _lib.lib.regnet_head_layer_train_supported(R, K, N):
            return False
        if not (bn.affine and bn.track_running_stats and bn.momentum is not None and conv.weight.is_cuda
                and tuple(conv.kernel_size) == (1,) and conv.groups == 1):
            return False
        # the kernels read the parameters and buffers as dense float32 rows (regnet_head_layer_train_fwd_f32 answers anything
        # else with ERR_SHAPE, which the trainer would record as a region error every step): anything else -> torch's layers
        tensors = (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        if not all(t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) for t in tensors):
            return False
        if conv.weight.data_ptr() % 16:
            return False
    return True


class _HeadTree(torch.autograd.Function):
    """forward(x (R, K0) contiguous, meta, *params) with params = (weight, bias, gamma, beta) per layer and meta = (tree,
    buffers): tree = ((parent, relu), ...), buffers = ((running_mean, running_var, num_batches_tracked, momentum, eps), ...).
    Returns the leaves' outputs (layers that are nobody's parent), in layer order."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        from . import _lib
        L = _lib.lib
        tree, buffers = meta
        nl = len(tree)
        R, K0 = x.shape
        dims = []                      # (K, N) per layer
        for l in range(nl):
            W = params[4 * l]
            dims.append((W.shape[1], W.shape[0]))
        is_parent = [False] * nl
        for parent, _ in tree:
            if parent >= 0:
                is_parent[parent] = True
        # one workspace: xhat of every layer, Y of the inner layers, invstd of every layer (offsets in floats, 16-byte steps)
        off, xh_off, y_off, inv_off = 0, [], [], []
        for l, (K, N) in enumerate(dims):
            xh_off.append(off); off += (R * N + 3) // 4 * 4
            if is_parent[l]:
                y_off.append(off); off += (R * N + 3) // 4 * 4
            else:
                y_off.append(-1)
            inv_off.append(off); off += (N + 3) // 4 * 4
        ws = torch.empty((off,), dtype=torch.float32, device=x.device)
        base = ws.data_ptr()
        leaves, y_ptr = [], []
        stream = _stream(x)
        with torch.cuda.device(x.device):
            for l, (K, N) in enumerate(dims):
                parent, relu = tree[l]
                W, b, gamma, beta = params[4 * l: 4 * l + 4]
                rm, rv, nbt, momentum, eps = buffers[l]
                if is_parent[l]:
                    yp = base + 4 * y_off[l]
                else:
                    out = torch.empty((R, N), dtype=torch.float32, device=x.device)
                    leaves.append(out)
                    yp = out.data_ptr()
                y_ptr.append(yp)
                src = x.data_ptr() if parent < 0 else y_ptr[parent]
                _lib.check(L.regnet_head_layer_train_fwd_f32(
                    src, K, W.data_ptr(), b.data_ptr() if b is not None else None, gamma.data_ptr(), beta.data_ptr(),
                    rm.data_ptr(), rv.data_ptr(), nbt.data_ptr() if nbt is not None else None, momentum, eps, R, K, N,
                    1 if relu else 0, base + 4 * xh_off[l], yp, base + 4 * inv_off[l], stream), "head_layer_train_fwd")
        ctx.save_for_backward(x, ws, *params)
        ctx.layout = (tree, dims, is_parent, xh_off, y_off, inv_off)
        CALLS["forward"] += 1
        return tuple(leaves)

    @staticmethod
    def backward(ctx, *grads):
        from . import _lib
        L = _lib.lib
        x, ws = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        tree, dims, is_parent, xh_off, y_off, inv_off = ctx.layout
        nl = len(tree)
        R = x.shape[0]
        dev = x.device
        base = ws.data_ptr()
        # gradient buffers: one flat tensor for dZ scratch (largest N) and the inner layers' dY; parameter gradients apart
        maxn = max(N for _, N in dims)
        goff, dy_off = (R * maxn + 3) // 4 * 4, {}
        for l, (K, N) in enumerate(dims):
            if is_parent[l]:
                dy_off[l] = goff; goff += (R * N + 3) // 4 * 4
        gws = torch.empty((goff,), dtype=torch.float32, device=dev)
        gbase = gws.data_ptr()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        leaf_grads = {}
        k = 0
        for l in range(nl):
            if not is_parent[l]:
                g = grads[k]; k += 1
                leaf_grads[l] = g.contiguous() if g is not None else torch.zeros((R, dims[l][1]), dtype=torch.float32, device=dev)
        pgrads = [None] * (4 * nl)
        written = set()               # inner layers (and -1 = x) whose gradient buffer already holds a branch's contribution
        stream = _stream(x)
        with torch.cuda.device(dev):
            for l in range(nl - 1, -1, -1):
                K, N = dims[l]
                parent, relu = tree[l]
                W, b, gamma, beta = params[4 * l: 4 * l + 4]
                dW = torch.empty_like(W)
                db = torch.empty((N,), dtype=torch.float32, device=dev) if b is not None else None
                dgamma = torch.empty((N,), dtype=torch.float32, device=dev)
                dbeta = torch.empty((N,), dtype=torch.float32, device=dev)
                if is_parent[l]:
                    dyp, yp = gbase + 4 * dy_off[l], base + 4 * y_off[l]
                else:
                    dyp, yp = leaf_grads[l].data_ptr(), None          # leaves have no ReLU: Y is not read
                    if relu:
                        raise RuntimeError("heads_train: a leaf layer with ReLU needs its output saved")
                if parent >= 0:
                    dxp, srcp = gbase + 4 * dy_off[parent], base + 4 * y_off[parent]
                else:
                    dxp, srcp = (dx.data_ptr() if dx is not None else None), x.data_ptr()
                _lib.check(L.regnet_head_layer_train_bwd_f32(
                    dyp, N, yp, base + 4 * xh_off[l], gamma.data_ptr(), base + 4 * inv_off[l], srcp, K, W.data_ptr(), R, K, N,
                    1 if relu else 0, gbase, dW.data_ptr(), db.data_ptr() if db is not None else None, dgamma.data_ptr(),
                    dbeta.data_ptr(), dxp, K, 1 if parent in written else 0, stream), "head_layer_train_bwd")
                written.add(parent)
                pgrads[4 * l: 4 * l + 4] = [dW, db, dgamma, dbeta]
        CALLS["backward"] += 1
        return (dx, None) + tuple(pgrads)


def _run(layers, x):
    x2 = x.reshape(x.shape[0], x.shape[1]).contiguous()
    if x2.data_ptr() % 16:      # an offset view that is already contiguous comes back unchanged: the kernels want 16-byte rows
        x2 = x2.clone()
    tree = tuple((parent, relu) for _, _, parent, relu in layers)
    buffers = tuple((bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum), float(bn.eps))
                    for _, bn, _, _ in layers)
    params = []
    for conv, bn, _, _ in layers:
        params += [conv.weight.view(conv.weight.shape[0], conv.weight.shape[1]), conv.bias, bn.weight, bn.bias]
    return _HeadTree.apply(x2, (tree, buffers), *params)


def twostage(module, mp_x):
    """PointNet2TwoStage's layers on the pooled rows mp_x (n, 256, 1) -> (x_cls (n, k_cls), x_reg (n, k_reg)) before the
    caller's view + sigmoid (pointnet2.py:184-188); check ``supported(_layers_twostage(module), mp_x)`` first."""
    return _run(_layers_twostage(module), mp_x)


def refine(module, x):
    """PointNet2Refine's layers on x (n, 384, 1) -> (x_cls (n, k_cls), x_reg (n, k_reg)) (pointnet2.py:246-253)."""
    return _run(_layers_refine(module), x)
