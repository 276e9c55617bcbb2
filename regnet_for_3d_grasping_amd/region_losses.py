"""The two grasp losses of the training iteration as a handful of launches each (csrc/losses.hip), with autograd.

What they replace: the label branches of ``GripperRegionNetwork.compute_loss`` / ``compute_loss_refine``
(multi_model/gripper_region_network.py:92-184, :233-309) -- ~160 / ~200 small tensor operations forward and as many again
backward on a few hundred rows, launch-bound and on the host-paced critical path of the iteration (DESIGN.md par. 12.5).  Same
formulas and the same consumption of numpy's global generator (the class-balancing draws stay on the host, in the reference's
order); values agree with the tensor code to fp32 rounding (tests/test_gpu_train.py).  GPU only; the tensor code remains the
path of CPU tensors (the oracle-backed mirror) and of ``FUSED = False``.
"""
import ctypes

import numpy as np
import torch

from . import _lib, host_io, np_random

FUSED = True
_L = _lib.lib
_check = _lib.check


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def usable(*tensors):
    return FUSED and all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


def usable_stage2(x_reg, x_cls, centres, ground):
    """The fused stage-2 loss is instantiated for the reference's configuration only (csrc/losses.hip:
    regnet_stage2_loss_rows_f32): 10 regression channels per anchor, label rows of at least 10 columns, at most 64 anchors.
    Anything else (``reg_channel`` is a constructor parameter) takes the tensor path, as the reference does."""
    return (usable(x_reg, x_cls, centres, ground) and x_reg.dim() == 3 and x_reg.shape[-1] == 10 and x_reg.shape[1] <= 64
            and ground.shape[-1] >= 10)


def usable_refine(next_grasp, next_x_cls, next_x_reg, next_gt):
    """Same for the refine loss (regnet_refine_loss_rows_f32: C == 10, label rows of >= 10 columns)."""
    return (usable(next_grasp, next_x_cls, next_x_reg, next_gt) and next_x_reg.dim() == 2 and next_x_reg.shape[-1] == 10
            and next_grasp.shape[-1] >= 10 and next_gt.shape[-1] >= 10)


class _Stage2Loss(torch.autograd.Function):
    """loss = 10 l_centre + 5 l_axis + l_theta + l_score + CE(class-balanced anchors); gradients to x_reg and x_cls."""

    @staticmethod
    def forward(ctx, x_reg, x_cls, centres, templates, labels, rows, radius):
        """x_reg (n,A,10), x_cls (n,A), centres (n, >= 3) rows, templates (A,4) float32, labels (n,10), rows (m) int64: the
        labelled centres.  -> (loss, values (12), next_grasp (m,10), pick (m), g8 (m), a_gt (m,7)); values = [l_centre, l_axis,
        l_theta, l_score, 4 monitoring terms of the arg-max decode, #(g8 == pick), #(g8 != pick), CE, 0]."""
        n, A, C = x_reg.shape
        m = int(rows.numel())
        if m == 0:
            # no labelled centre in the batch: the tensor path fails in ``np.concatenate([])`` (ValueError) and the reference's
            # bare ``except`` (train.py:430) falls back to the ScoreNet loss alone -- same exception type here, raised before
            # anything is launched or drawn from numpy's stream
            raise ValueError("stage-2 loss: no labelled centre in the batch")
        dev = x_reg.device
        x_reg, x_cls, labels = x_reg.contiguous(), x_cls.contiguous(), labels.contiguous()
        centres = centres if centres.stride(1) == 1 else centres.contiguous()
        weights = (ctypes.c_float * 4)(10.0 / (3 * m), 5.0 / (3 * m), 1.0 / m, 1.0 / (3 * m))
        with torch.cuda.device(dev):
            next_grasp = torch.empty((m, C), dtype=torch.float32, device=dev)
            pick = torch.empty((m,), dtype=torch.int32, device=dev)
            g8 = torch.empty((m,), dtype=torch.int32, device=dev)
            a_gt = torch.empty((m, 7), dtype=torch.float32, device=dev)
            terms = torch.empty((m, 12), dtype=torch.float32, device=dev)
            dreg = torch.zeros((n, A, C), dtype=torch.float32, device=dev)
            dcls = torch.zeros((n, A), dtype=torch.float32, device=dev)
            _check(_L.regnet_stage2_loss_rows_f32(x_cls.data_ptr(), x_reg.data_ptr(), A, C, centres.data_ptr(),
                                                  centres.stride(0), templates.data_ptr(), labels.data_ptr(),
                                                  labels.stride(0), float(radius), ctypes.addressof(weights), rows.data_ptr(),
                                                  m, next_grasp.data_ptr(), pick.data_ptr(), g8.data_ptr(), a_gt.data_ptr(),
                                                  terms.data_ptr(), dreg.data_ptr(), _stream(x_reg)), "stage2_loss_rows")
            # class-balanced subset: the same number of centres per (non-empty) anchor class, drawn from numpy's global stream
            # in class order (gripper_region_network.py:108-130) -- one device->host read of the <= B*64 class ids
            classes = g8.cpu().numpy()
            members = [np.nonzero(classes == a)[0] for a in range(A)]
            per_class = max(int(min(len(mem) for mem in members)), 1)
            np_random.flush()
            chosen = [mem[np.random.choice(len(mem), per_class, replace=False)] for mem in members if len(mem)]
            # (the picks and the two constant vectors of the reduction below in one transfer)
            idx, scale, mix = host_io.upload_many((
                np.concatenate(chosen).astype(np.int64),
                np.array([1.0 / (3 * m), 1.0 / (3 * m), 1.0 / m, 1.0 / (3 * m), 1.0 / (3 * m), 1.0 / m, 1.0 / m, 1.0 / (3 * m), 1.0,
                          0.0, 0.0, 0.0], dtype=np.float32),
                np.array([10.0, 5.0, 1.0, 1.0, 0, 0, 0, 0, 0, 0, 1.0, 0], dtype=np.float32)), dev)
            nb = int(idx.numel())
            ce_rows = torch.empty((nb,), dtype=torch.float32, device=dev)
            _check(_L.regnet_ce_rows_f32(x_cls.data_ptr(), A, g8.data_ptr(), idx.data_ptr(), rows.data_ptr(), nb, 1.0 / nb,
                                         ce_rows.data_ptr(), dcls.data_ptr(), _stream(x_reg)), "ce_rows")
            sums = terms.sum(0)
            ce = ce_rows.sum()
            values = sums * scale
            values[9] = m - values[8]
            values[10] = ce / nb
            loss = torch.dot(values, mix)
        ctx.save_for_backward(dreg, dcls)
        ctx.mark_non_differentiable(values, next_grasp, pick, g8, a_gt)
        return loss, values, next_grasp, pick, g8, a_gt

    @staticmethod
    def backward(ctx, g_loss, *unused):
        dreg, dcls = ctx.saved_tensors
        return dreg * g_loss, dcls * g_loss, None, None, None, None, None


def stage2_loss(x_reg, x_cls, centres, templates, ground, radius):
    """The label branch of ``GripperRegionNetwork.compute_loss``: -> (next_grasp, loss_tuple, correct_tuple, next_gt, a_gt,
    gmask) exactly as that method returns them."""
    labels = ground.view(-1, ground.shape[2])
    gmask = torch.nonzero(labels[:, -1] != -1).view(-1)
    loss, v, next_grasp, _, _, a_gt = _Stage2Loss.apply(x_reg, x_cls, centres, templates, labels, gmask, radius)
    loss_tuple = (loss, v[10], v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
    correct_tuple = (v[8], v[9])
    next_gt = labels[gmask]
    return next_grasp, loss_tuple, correct_tuple, next_gt, a_gt, gmask


class _RefineLoss(torch.autograd.Function):
    """loss = CE(class-balanced label-positive / negative rows) + four smooth-L1 terms on the label-positive rows."""

    @staticmethod
    def forward(ctx, next_x_reg, next_x_cls, next_grasp, next_gt, radius, score_thre):
        m, C = next_x_reg.shape
        dev = next_x_reg.device
        next_x_reg, next_x_cls = next_x_reg.contiguous(), next_x_cls.contiguous()
        grasp = next_grasp if next_grasp.stride(1) == 1 else next_grasp.contiguous()
        gt = next_gt if next_gt.stride(1) == 1 else next_gt.contiguous()
        with torch.cuda.device(dev):
            final = torch.empty((m, C), dtype=torch.float32, device=dev)
            flags8 = torch.empty((3, m), dtype=torch.uint8, device=dev)
            terms = torch.empty((m, 20), dtype=torch.float32, device=dev)
            dreg = torch.empty((m, C), dtype=torch.float32, device=dev)
            dcls = torch.zeros((m, 2), dtype=torch.float32, device=dev)
            _check(_L.regnet_refine_loss_rows_f32(grasp.data_ptr(), grasp.stride(0), next_x_cls.data_ptr(),
                                                  next_x_reg.data_ptr(), gt.data_ptr(), gt.stride(0), C, float(radius),
                                                  float(score_thre), m, final.data_ptr(), flags8.data_ptr(), terms.data_ptr(),
                                                  dreg.data_ptr(), _stream(next_x_reg)), "refine_loss_rows")
            flags = flags8.cpu().numpy().astype(bool)                  # the one read: class / score / label-positive flags
            class_np, score_np = np.nonzero(flags[0])[0], np.nonzero(flags[1])[0]
            pos_np, neg_np = np.nonzero(flags[2])[0], np.nonzero(~flags[2])[0]
            num = min(len(neg_np), len(pos_np))
            P, nc, ns = len(pos_np), len(class_np), len(score_np)
            sums = terms.sum(0)
            nan = float("nan")
            reg_scale = [1.0 / (3 * P), 1.0 / (3 * P), 1.0 / P, 1.0 / (3 * P)] if num > 0 else [0.0] * 4
            if nc > 0:      # (the reference computes the score-kept terms whenever a class-1 row exists: empty means are nan)
                mon = ([1.0 / (3 * nc), 1.0 / nc, 1.0 / nc, 1.0 / (3 * nc)] * 2
                       + ([1.0 / (3 * ns), 1.0 / ns, 1.0 / ns, 1.0 / (3 * ns)] if ns > 0 else [nan] * 4))
            else:
                mon = [0.0] * 12
            # every host array of this loss in ONE transfer: the scale vector, both selections and -- with a class-balanced subset
            # to draw (numpy's stream, as gripper_region_network.py:262-268) -- its rows and the gradient's column scale
            hosts = [np.array(reg_scale + mon + [1.0] * 4, dtype=np.float32), class_np, score_np]
            if num > 0:
                np_random.flush()
                idx0 = neg_np[np.random.choice(len(neg_np), num, replace=False)]
                idx1 = pos_np[np.random.choice(len(pos_np), num, replace=False)]
                hosts += [np.concatenate((idx0, idx1)).astype(np.int64),
                          np.array(reg_scale[:1] * 3 + reg_scale[1:2] * 3 + reg_scale[2:3] + reg_scale[3:4] * 3, dtype=np.float32)]
            up = host_io.upload_many(hosts, dev)
            scale, class_t, score_t = up[0], up[1], up[2]
            values = sums * scale
            if nc > 0 and ns == 0:
                values[12:16] = nan                                    # 0 * nan above is nan already; written for clarity
            ce = torch.zeros((), dtype=torch.float32, device=dev)
            if num > 0:
                idx, col = up[3], up[4]
                nb = int(idx.numel())
                target = flags8[2].to(torch.int32)
                ce_rows = torch.empty((nb,), dtype=torch.float32, device=dev)
                _check(_L.regnet_ce_rows_f32(next_x_cls.data_ptr(), 2, target.data_ptr(), idx.data_ptr(), None, nb, 1.0 / nb,
                                             ce_rows.data_ptr(), dcls.data_ptr(), _stream(next_x_reg)), "ce_rows")
                ce = ce_rows.sum() / nb
                dreg = dreg * col
                loss = ce + values[:4].sum()
            else:
                dreg = torch.zeros_like(dreg)
                loss = torch.zeros((), dtype=torch.float32, device=dev)
        ctx.save_for_backward(dreg, dcls)
        ctx.has_loss = num > 0
        extras = (values, ce, final, class_t, score_t)
        ctx.mark_non_differentiable(*extras)
        return (loss,) + extras

    @staticmethod
    def backward(ctx, g_loss, *unused):
        dreg, dcls = ctx.saved_tensors
        if not ctx.has_loss:
            return None, None, None, None, None, None
        return dreg * g_loss, dcls * g_loss, None, None, None, None


def refine_loss(next_grasp, next_x_cls, next_x_reg, next_gt, radius, score_thre):
    """The label branch of ``GripperRegionNetwork.compute_loss_refine``: -> (sel_class, sel_score, sel_class_stage2,
    class_select, score_select, loss_refine_tuple, counts) exactly as that method returns them."""
    loss, v, ce, final, class_select, score_select = _RefineLoss.apply(next_x_reg, next_x_cls, next_grasp, next_gt, radius,
                                                                       score_thre)
    sel_class, sel_score = final[class_select], final[score_select]
    sel_class_stage2 = next_grasp[class_select].data
    loss_refine_tuple = (loss, ce, v[0], v[1], v[2], v[3]) + tuple(v[4 + i] for i in range(12))
    counts = (v[16], v[17], v[18], v[19])
    return sel_class, sel_score, sel_class_stage2, class_select, score_select, loss_refine_tuple, counts
