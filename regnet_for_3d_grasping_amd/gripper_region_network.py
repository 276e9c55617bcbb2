"""Grasp-region network + refine stage, forward path (mirror of
multi_model/gripper_region_network.py:10-44, :361-434, :311-359, :436-610).

Same class / function names, constructor and ``forward`` signatures and the same 16-tuple
result as the reference.  The data-parallel pieces run on MI355X kernels:
  * grouped-feature gather + MaxPool1d  -> ``region_ops.gather_max``   (reference :388-395, :334-343)
  * gripper closing-box membership      -> ``region_ops.box_candidates`` (reference :508-544)
while the host only draws numpy's random positions in the reference's call order.

Reference quirks that are reproduced on purpose (SURVEY.md §7 "hard parts"):
  * anchor templates are rounded through fp16 (``.half()``, :586);
  * the refine stage re-views the pooled (n,256,1) region feature as 128-wide rows (:343);
  * ``gripper_pc`` / index tensors are created by ``torch.full(..., -1)`` as int64 (:517-520).

The training losses (``ground_grasp`` given; :92-184, :233-309) are plain torch code (not kernel
targets); they draw their class-balancing samples from numpy's global RNG like the reference.
"""
import math
import threading

import numpy as np
import torch
from torch import nn

from . import host_io, np_random, region_losses, region_ops
from .pointnet2 import PointNet2Refine, PointNet2TwoStage


def _pool_rows(all_feature, rows):
    """max over the G gathered rows: all_feature (B,N,F), rows (R,G) global row ids -> (R,F,1)."""
    F = all_feature.shape[2]
    if all_feature.is_cuda and torch.is_grad_enabled() and all_feature.requires_grad and all_feature.dim() == 3:
        # training: gather from a copy made WITHOUT autograd; the pool's own node hands the gradient to the map (channel-first)
        return region_ops.gather_max_map_train(all_feature, _contiguous_rows(all_feature, detach=True), rows).unsqueeze(-1)
    flat = _contiguous_rows(all_feature)
    if flat.is_cuda and not torch.is_grad_enabled():
        return region_ops.gather_max(flat, rows).unsqueeze(-1)
    if flat.is_cuda and hasattr(region_ops, "gather_max_train"):
        return region_ops.gather_max_train(flat, rows).unsqueeze(-1)     # same values; the backward scatters R x F values
    # autograd path: the reference's materialised gather followed by a max over the group axis
    return flat[rows.reshape(-1)].view(rows.shape[0], rows.shape[1], F).max(dim=1)[0].unsqueeze(-1)


def _scene_pool(all_feature, index):
    """True when the pooling kernel can address by scene (region_ops.gather_max_scene): eval on the GPU, 16-byte feature
    rows, int64 local indices."""
    return (all_feature.is_cuda and not torch.is_grad_enabled() and all_feature.dim() == 3 and all_feature.shape[2] % 4 == 0
            and all_feature.dtype == torch.float32 and index.dtype == torch.int64 and index.is_cuda)


class _RowsCache(threading.local):
    """Per THREAD (a training call and an inference call on two threads must not see each other's copy)."""
    ref = None
    flat = None


_rows_cache = _RowsCache()


def forget_rows():
    """Drop the cached contiguous rows (the copy is 210 MB at B = 8; in training it is held strongly from the region head's
    pool to the refine head's).  ``GripperRegionNetwork.forward`` calls this when it returns, so the copy never outlives the
    forward that used it, whoever drives the network; ``RefineTrainer.step`` calls it again at the end of the iteration."""
    _rows_cache.ref = _rows_cache.flat = None


def _contiguous_rows(all_feature, detach=False):
    """``all_feature.contiguous().view(-1, F)``, made ONCE per feature map: ScoreNet hands the map out as a transposed view
    in training, the region head and the refine head both pool from it, and every ``.contiguous()`` is a 210 MB transpose
    copy (B = 8).  Keyed on the tensor object, weakly.  ``detach``: the copy carries no graph (GPU training: the pools
    route their gradient themselves, region_ops._GatherMaxMapFn) and is held strongly until ``forget_rows``; otherwise the
    copy is held only while it requires grad, never in eval mode."""
    import weakref
    ref, flat = _rows_cache.ref, _rows_cache.flat
    if ref is not None and ref() is all_feature and flat is not None and (not detach or not flat.requires_grad):
        return flat
    src = all_feature.detach() if detach else all_feature
    flat = src.contiguous().view(-1, all_feature.shape[2])
    _rows_cache.ref, _rows_cache.flat = weakref.ref(all_feature), (flat if (detach or flat.requires_grad) else None)
    return flat


class GripperRegionNetwork(nn.Module):
    def __init__(self, training, group_num, gripper_num, grasp_score_threshold, radius, reg_channel):
        super().__init__()
        self.group_number = group_num
        self.templates = _enumerate_templates()
        self.anchor_number = self.templates.shape[1] * self.templates.shape[2]
        self.gripper_number = gripper_num
        self.grasp_score_thre = grasp_score_threshold
        self.is_training_refine = training
        self.radius = radius
        self.reg_channel = reg_channel

        self.extrat_feature_region = PointNet2TwoStage(num_points=group_num, input_chann=6,
                                                       k_cls=self.anchor_number,
                                                       k_reg=self.reg_channel * self.anchor_number,
                                                       k_reg_theta=self.anchor_number)
        self.extrat_feature_refine = PointNet2Refine(num_points=gripper_num, input_chann=6, k_cls=2,
                                                     k_reg=self.reg_channel)
        self.criterion_cos = nn.CosineEmbeddingLoss(reduction="mean")
        self.criterion_cls = nn.CrossEntropyLoss(reduction="mean")
        self.smooth_l1_loss = nn.SmoothL1Loss(reduction="mean")

    def _enumerate_anchors(self, centers):
        """centers (n,3) -> anchors (n, A, 7) = [centre xyz | template r (3) | template theta]
        (gripper_region_network.py:30-44)."""
        self.templates = self.templates.to(centers.device)
        n = centers.shape[0]
        tmpl = self.templates.float().view(1, self.anchor_number, 4).expand(n, -1, -1)
        return torch.cat([centers.view(n, 1, 3).expand(-1, self.anchor_number, -1), tmpl], dim=-1)

    def _decode(self, grasp, anchor):
        """(delta, anchor) -> grasp: centre = delta*radius + anchor, closing axis re-normalised,
        theta = pi*(delta + anchor), remaining channels passed through (:76-90)."""
        axis = grasp[:, 3:6] + anchor[:, 3:6]
        norm = torch.sqrt(torch.sum(torch.mul(axis, axis), dim=1).add_(1e-12)).view(-1, 1)
        return (grasp[:, :3] * self.radius + anchor[:, :3], torch.div(axis, norm),
                np.pi * (grasp[:, 6:7] + anchor[:, 6:7]), grasp[:, 7:], norm)

    def compute_loss(self, first_grasp, anchors, first_cls, ground):
        """Stage-2 decode (+ loss when ``ground`` (B,Nc,10) labels are given)
        (gripper_region_network.py:46-184).

        Decode: the regression of the arg-max anchor of every labelled centre becomes ``next_grasp``.
        Loss: anchors are classified against the template whose axis is closest (cosine) to the
        label's, with class-balanced sampling (numpy RNG); the regression of THAT anchor is trained
        with smooth-L1 terms: 10*centre + 5*(delta_r*|r|) + theta + score + CE."""
        n = first_grasp.shape[0]
        dev = first_grasp.device
        if ground is not None:
            gmask = torch.nonzero(ground.view(-1, ground.shape[2])[:, -1] != -1).view(-1).to(dev)
        else:
            gmask = torch.arange(0, n, device=dev)
        if ground is not None:      # (without labels gmask is the identity: no gathers)
            anchors, first_grasp, first_cls = anchors[gmask], first_grasp[gmask], first_cls[gmask]
        anchors = anchors.detach()
        m, A = first_cls.shape
        rows = torch.arange(m, device=dev)

        pick = torch.max(first_cls, dim=1)[1]
        g_pre, a_pre = first_grasp[rows, pick], anchors[rows, pick]
        center_pre, r_pre, angle_pre, score_pre, _ = self._decode(g_pre, a_pre)
        next_grasp = torch.cat((center_pre, r_pre, angle_pre, score_pre), dim=-1)
        if ground is None:
            return next_grasp, (None, None), (None, None, None, None), None, None, gmask

        labels = ground.view(-1, ground.shape[2])[gmask]
        gt7, gt_score = labels[:, :7], labels[:, 7:]
        # anchor whose orientation template is most similar to the label's closing axis
        sim = torch.stack([compute_cos_sim(anchors[:, a, 3:6], gt7[:, 3:6]).view(-1) for a in range(A)], dim=1)
        ground_8 = torch.sort(sim, dim=1, descending=False)[1][:, 0]

        # class-balanced subset: the same number of centres per (non-empty) anchor class.  ONE device->host read (the
        # <= B*64 class ids) instead of a count + a nonzero per class; the draws stay on numpy's global stream, in class order
        classes = ground_8.cpu().numpy()
        members = [np.nonzero(classes == a)[0] for a in range(A)]
        per_class = max(int(min(len(mem) for mem in members)), 1)
        np_random.flush()    # host-side draws below: numpy's generator must hold the state the device draws left
        chosen = [mem[np.random.choice(len(mem), per_class, replace=False)] for mem in members if len(mem)]
        balanced = host_io.upload(np.concatenate(chosen).astype(np.int64), dev)
        loss_class = self.criterion_cls(first_cls[balanced], ground_8[balanced].long())
        correct_tuple = ((ground_8 == pick).sum().float(), (ground_8 != pick).sum().float())

        g_gt, a_gt = first_grasp[rows, ground_8], anchors[rows, ground_8]
        center_gt, r_gt, angle_gt, score_gt, norm_gt = self._decode(g_gt, a_gt)
        sl1 = nn.functional.smooth_l1_loss
        l_center = sl1(g_gt[:, :3], (gt7[:, :3] - a_gt[:, :3]) / self.radius, reduction="mean")
        l_axis = sl1(torch.mul(g_gt[:, 3:6], norm_gt), gt7[:, 3:6] - a_gt[:, 3:6], reduction="mean")
        l_theta = sl1(g_gt[:, 6:7], (gt7[:, 6:7] - a_gt[:, 6:7]) / np.pi, reduction="mean")
        l_score = sl1(g_gt[:, 7:], gt_score, reduction="mean")

        # all-ones target; the reference passes an (m,1) tensor, which its torch 1.8 broadcast to the same value
        ones = torch.ones(m, device=dev)
        with torch.no_grad():  # monitoring terms of the arg-max ("pre") decode, never back-propagated (.data in the reference)
            mon = (sl1(center_pre, gt7[:, :3], reduction="mean"), self.criterion_cos(r_pre, gt7[:, 3:6], ones),
                   sl1(angle_pre, gt7[:, 6:7], reduction="mean"), sl1(score_pre, gt_score, reduction="mean"))
        next_gt = torch.cat((gt7, gt_score), dim=1)
        loss = l_center * 10 + l_axis * 5 + l_theta + l_score + loss_class
        loss_tuple = (loss, loss_class.data, l_center.data, l_axis.data, l_theta.data, l_score.data) + mon
        return next_grasp, loss_tuple, correct_tuple, next_gt, a_gt, gmask

    def compute_loss_refine(self, next_grasp, next_x_cls, next_x_reg, next_gt):
        """Apply the refine deltas, select class-1 grasps, and -- with labels -- the refine loss
        (gripper_region_network.py:186-309): a grasp is a positive when its stage-2 centre is within
        2.5 cm, its axis within 60 deg (1 - cos < 0.5) and its angle within 1.047 rad of the label;
        CE on a class-balanced subset (numpy RNG) + four smooth-L1 terms on the positives."""
        dev = next_grasp.device
        if next_gt is not None and region_losses.usable_refine(next_grasp, next_x_cls, next_x_reg, next_gt):
            return region_losses.refine_loss(next_grasp, next_x_cls, next_x_reg, next_gt, self.radius, self.grasp_score_thre)
        if (next_gt is None and next_grasp.is_cuda and not torch.is_grad_enabled() and next_grasp.dtype == torch.float32
                and hasattr(region_ops, "refine_decode")):
            # inference: deltas, class arg-max and both selections' flags in ONE launch, one read (csrc/region.hip)
            final_grasp, flags8 = region_ops.refine_decode(next_grasp, next_x_cls, next_x_reg, self.radius,
                                                           self.grasp_score_thre)
            flags = flags8.cpu().numpy().astype(bool)
            class_select, score_select = host_io.upload_many((np.nonzero(flags[0])[0], np.nonzero(flags[1])[0]), dev)
            return (final_grasp[class_select], final_grasp[score_select], next_grasp[class_select], class_select,
                    score_select, (None, None), (None, None, None, None))
        final_grasp = next_grasp.clone()
        final_grasp[:, :3] = final_grasp[:, :3] + next_x_reg[:, :3] * self.radius
        final_grasp[:, 3:] = final_grasp[:, 3:] + next_x_reg[:, 3:]
        predicted = torch.max(next_x_cls, dim=-1)[1]
        is_class = predicted == 1
        is_score = is_class & (final_grasp[:, 7] > self.grasp_score_thre)
        if next_gt is None:
            flags = torch.stack((is_class, is_score)).cpu().numpy()                      # one read for both selections
        else:
            offset = next_grasp[:, :3] - next_gt[:, :3]
            near = torch.sqrt(offset[:, 0] * offset[:, 0] + offset[:, 1] * offset[:, 1] + offset[:, 2] * offset[:, 2]) < 0.025
            aligned = compute_cos_sim(next_grasp[:, 3:6], next_gt[:, 3:6]).view(-1) < 0.5
            same_angle = torch.abs(next_grasp[:, 6] - next_gt[:, 6]) < 1.047
            gt_positive = near & aligned & same_angle
            flags = torch.stack((is_class, is_score, gt_positive)).cpu().numpy()         # ... and the label classes
        class_select, score_select = host_io.upload_many((np.nonzero(flags[0])[0], np.nonzero(flags[1])[0]), dev)
        sel_class, sel_score = final_grasp[class_select].data, final_grasp[score_select].data
        sel_class_stage2 = next_grasp[class_select].data
        if next_gt is None:
            return (sel_class, sel_score, sel_class_stage2, class_select, score_select, (None, None),
                    (None, None, None, None))

        gt_class = gt_positive.float()
        pos_np, neg_np = np.nonzero(flags[2])[0], np.nonzero(~flags[2])[0]
        pos, neg = host_io.upload_many((pos_np, neg_np), dev)
        num = min(len(neg_np), len(pos_np))

        zero = torch.zeros((), device=dev)
        loss = loss_class = l_center = l_axis = l_theta = l_score = zero
        sl1 = nn.functional.smooth_l1_loss
        if num > 0:
            np_random.flush()
            idx0 = neg_np[np.random.choice(len(neg_np), num, replace=False)]
            idx1 = pos_np[np.random.choice(len(pos_np), num, replace=False)]
            index = host_io.upload(np.concatenate((idx0, idx1)).astype(np.int64), dev)
            loss_class = self.criterion_cls(next_x_cls.view(-1, 2)[index], gt_class.view(-1)[index].long())
            l_center = sl1(next_x_reg[pos, :3], (next_gt[pos, :3] - next_grasp[pos, :3]) / self.radius, reduction="mean")
            l_axis = sl1(next_x_reg[pos, 3:6], next_gt[pos, 3:6] - next_grasp[pos, 3:6], reduction="mean")
            l_theta = sl1(next_x_reg[pos, 6], next_gt[pos, 6] - next_grasp[pos, 6], reduction="mean")
            l_score = sl1(next_x_reg[pos, 7:], next_gt[pos, 7:] - next_grasp[pos, 7:], reduction="mean")
            loss = loss_class + l_center + l_axis + l_theta + l_score

        mon = [zero] * 12   # stage2 | stage3-class | stage3-score monitoring terms (centre, axis, theta, score)
        if len(class_select) > 0:
            with torch.no_grad():
                def terms(pred, sel):
                    gt = next_gt[sel]
                    return [sl1(pred[:, :3], gt[:, :3], reduction="mean"),
                            self.criterion_cos(pred[:, 3:6], gt[:, 3:6], torch.ones(len(pred), device=dev)),
                            sl1(pred[:, 6], gt[:, 6], reduction="mean"), sl1(pred[:, 7:], gt[:, 7:], reduction="mean")]
                mon = terms(sel_class_stage2, class_select) + terms(sel_class, class_select) + \
                    terms(sel_score, score_select)
        flat_pred = predicted.view(-1)
        counts = tuple(((gt_class == g) & (flat_pred == p)).sum().float() for g, p in ((1, 1), (0, 0), (0, 1), (1, 0)))
        loss_refine_tuple = (loss, loss_class.data, l_center.data, l_axis.data, l_theta.data, l_score) + tuple(mon)
        return sel_class, sel_score, sel_class_stage2, class_select, score_select, loss_refine_tuple, counts

    def refine_forward(self, pc_group_more_xyz, pc_group_more_index, true_mask, all_feature, group_feature_mp,
                       next_grasp, gripper_params, next_gt=None, all_kept=False):
        """Crop the gripper closing box out of the large groups, pool the ScoreNet features of
        the cropped points and refine the grasps (gripper_region_network.py:311-359).  ``all_kept``: ``true_mask`` is
        every centre in order (inference without labels), so the reference's ``[true_mask]`` gathers are identities and
        are not made (12.6 + 4.2 MB of copies per batch of 8)."""
        B, N = all_feature.shape[0], all_feature.shape[1]
        N_C, N_G_M = pc_group_more_index.shape[1], pc_group_more_index.shape[2]
        if all_kept:
            group_xyz, group_index = pc_group_more_xyz, pc_group_more_index.view(-1, N_G_M)
        else:
            group_xyz, group_index = pc_group_more_xyz[true_mask], pc_group_more_index.view(-1, N_G_M)[true_mask]
        _, _, index_inall, gripper_mask = get_gripper_region_transform(
            group_xyz, group_index, next_grasp, self.gripper_number, gripper_params, points_too=False)
        out = [None, None, None, None, None, (None, None), (None, None), next_gt]
        self.last_valid_crops = int(len(gripper_mask))    # rows of the refine network in this call (host-known, no sync)
        if len(gripper_mask) >= 2:
            if all_kept and _scene_pool(all_feature, index_inall) and gripper_mask.dtype == torch.int64:
                scene_off = None
            elif all_kept:
                scene_off = _scene_offsets(B, N_C, N, true_mask.device)
            else:
                scene = torch.arange(B, device=true_mask.device).view(-1, 1).repeat(1, N_C).view(-1)[true_mask]
                scene_off = scene.view(-1, 1) * N
            if all_kept and _scene_pool(all_feature, index_inall) and gripper_mask.dtype == torch.int64:
                gripper_feature = region_ops.gather_max_scene(_contiguous_rows(all_feature), index_inall, gripper_mask, N_C,
                                                              N).unsqueeze(-1)
            else:
                rows = (index_inall.long() + scene_off)[gripper_mask]
                gripper_feature = _pool_rows(all_feature, rows)                   # (m, F, 1)
            region_feature = group_feature_mp.view(-1, 128)[gripper_mask].contiguous()  # the 128-wide re-view quirk
            next_x_cls, next_x_reg = self.extrat_feature_refine(gripper_feature, region_feature, pooled=True)
            if next_gt is not None:
                next_gt = next_gt[gripper_mask]
            (out[0], out[1], out[2], class_select, score_select, out[5], out[6]) = self.compute_loss_refine(
                next_grasp[gripper_mask], next_x_cls, next_x_reg, next_gt)
            if next_gt is not None:
                next_gt = next_gt[class_select]
            kept = true_mask.clone()[gripper_mask]
            out[3], out[4], out[7] = kept[class_select], kept[score_select], next_gt
        return tuple(out)

    def forward(self, pc_group, pc_group_more, pc_group_index, pc_group_more_index, center_pc, center_pc_index, pc,
                all_feature, gripper_params, ground_grasp=None, data_path=None):
        """Shapes as the reference (gripper_region_network.py:361-375); returns its 16-tuple."""
        try:
            return self._forward(pc_group, pc_group_more, pc_group_index, pc_group_more_index, center_pc, center_pc_index, pc,
                                 all_feature, gripper_params, ground_grasp, data_path)
        finally:
            forget_rows()     # the contiguous copy of the feature map both pools gathered from (their autograd nodes keep
                              # indices, not the copy)

    def _forward(self, pc_group, pc_group_more, pc_group_index, pc_group_more_index, center_pc, center_pc_index, pc,
                 all_feature, gripper_params, ground_grasp=None, data_path=None):
        B, N_C, N_G, _ = pc_group.shape
        N = all_feature.shape[1]
        large_groups = pc_group_more_index if callable(pc_group_more_index) else None   # get_regiondataset.DEFER_LARGE_GROUPS
        if large_groups is None:
            pc_group_more_xyz = pc_group_more[:, :, :, :6].reshape(B * N_C, -1, 6)

        if _scene_pool(all_feature, pc_group_index):
            # inference on the GPU: the reference's `index + b * N` is formed in the pooling kernel's address, not as a tensor
            pooled = region_ops.gather_max_scene(_contiguous_rows(all_feature), pc_group_index.view(B * N_C, N_G), None,
                                                 N_C, N).unsqueeze(-1)
        else:
            rows = (pc_group_index.long().view(B, N_C * N_G) + _scene_offsets(B, 1, N, pc_group_index.device)).view(B * N_C, N_G)
            pooled = _pool_rows(all_feature, rows)                                # (B*N_C, F, 1)
        # inference without labels on the GPU: the head hands its regression over raw and ONE kernel decodes the arg-max
        # anchor of every centre (region_ops.stage2_decode) -- no anchor tensor, no gathers; every centre is kept
        fast = (ground_grasp is None and pooled.is_cuda and not torch.is_grad_enabled() and center_pc.dtype == torch.float32
                and hasattr(region_ops, "stage2_decode"))
        x_cls, x_reg, mp_center_feature = self.extrat_feature_region(pooled, None, pooled=True, raw_reg=fast)
        if large_groups is not None:
            # the large groups' draws (next on numpy's stream, before the loss's class-balancing draws) and their resampling,
            # made by the host while the device is busy with the pool and the head just enqueued
            pc_group_more_index, pc_group_more = large_groups()
            pc_group_more_xyz = pc_group_more[:, :, :, :6].reshape(B * N_C, -1, 6)
        if fast:
            self.templates = self.templates.to(center_pc.device)
            tmpl = _float_templates(self.templates)
            next_grasp = region_ops.stage2_decode(x_cls, x_reg, center_pc.reshape(B * N_C, -1), tmpl, self.radius,
                                                  self.extrat_feature_region.reg_is_raw)
            true_mask, keep2 = _all_centres(B, N_C, center_pc.device)
            loss_tuple, correct_tuple, next_gt = (None, None), (None, None, None, None), None
        elif (ground_grasp is not None and center_pc.dtype == torch.float32
              and region_losses.usable_stage2(x_reg, x_cls, center_pc, ground_grasp)):
            # training on the GPU: the whole label branch of compute_loss (decode, anchor matching, four smooth-L1 terms, the
            # class-balanced cross entropy, monitoring) in two launches + one read, gradients included (csrc/losses.hip)
            self.templates = self.templates.to(center_pc.device)
            next_grasp, loss_tuple, correct_tuple, next_gt, _, true_mask = region_losses.stage2_loss(
                x_reg, x_cls, center_pc.reshape(B * N_C, -1), _float_templates(self.templates), ground_grasp, self.radius)
            keep2 = _per_scene_counts(true_mask, N_C, B)
        else:
            anchors = self._enumerate_anchors(center_pc[:, :, :3].reshape(-1, 3).float())
            next_grasp, loss_tuple, correct_tuple, next_gt, _, true_mask = self.compute_loss(x_reg, anchors, x_cls,
                                                                                            ground_grasp)
            keep2 = _per_scene_counts(true_mask, N_C, B)

        res = (None,) * 5 + (None, None, None)
        keep3 = keep3_score = None
        if self.is_training_refine:
            res = self.refine_forward(pc_group_more_xyz, pc_group_more_index, true_mask, all_feature,
                                      mp_center_feature, next_grasp.detach(), gripper_params, next_gt, all_kept=fast)
            final_mask, final_mask_sthre = res[3], res[4]
            if final_mask is not None:
                keep3 = _per_scene_counts(final_mask, N_C, B)
                keep3_score = _per_scene_counts(final_mask_sthre, N_C, B)
            else:
                keep3, keep3_score = [0] * B, [0] * B
        (select_class, select_score, select_class_stage2, final_mask, final_mask_sthre, loss_refine_tuple,
         correct_refine_tuple, gt) = res
        np_random.flush_unless_deferred()   # hand numpy's generator the state the crop draws left on the device
        return (next_grasp.detach(), keep2, true_mask, loss_tuple, correct_tuple, next_gt, select_class,
                select_score, select_class_stage2, keep3, keep3_score, final_mask, final_mask_sthre,
                loss_refine_tuple, correct_refine_tuple, gt)


_template_cache = {}


def _float_templates(templates):
    """The fp16 anchor templates as a contiguous (A,4) float32 tensor on their device, converted once per tensor object."""
    key = (id(templates), templates.device)
    hit = _template_cache.get(key)
    if hit is None or hit[0] is not templates:
        _template_cache.clear()
        hit = (templates, templates.float().reshape(-1, 4).contiguous())
        _template_cache[key] = hit
    return hit[1]


_centre_cache = {}
_offset_cache = {}


def _scene_offsets(B, per_scene, N, device):
    """(B * per_scene, 1) int64: scene index of every centre times N (the row offset of its scene in the flattened feature
    map), built once per (B, per_scene, N, device); read-only."""
    key = (B, per_scene, N, device)
    if key not in _offset_cache:
        _offset_cache[key] = (torch.arange(B * per_scene, device=device) // per_scene * N).view(-1, 1)
    return _offset_cache[key]



def _all_centres(B, per_scene, device):
    """(true_mask, keep counts) when every centre is kept (no labels): arange(B * per_scene) and B 0-dim tensors holding
    ``per_scene``, built once per (B, per_scene, device) -- read-only results (callers index with them, never write)."""
    key = (B, per_scene, device)
    if key not in _centre_cache:
        _centre_cache[key] = (torch.arange(0, B * per_scene, device=device),
                              list(torch.full((B,), per_scene, dtype=torch.int64, device=device).unbind(0)))
    mask, keep = _centre_cache[key]
    return mask, list(keep)


def _per_scene_counts(ids, per_scene, B):
    """[#ids falling into scene i's range [i*per_scene, (i+1)*per_scene) for i in range(B)] as 0-dim
    tensors (the reference builds the same list with B masked sums, :411, :424-425)."""
    scene = torch.div(ids, per_scene, rounding_mode="floor").view(-1, 1)
    return list((scene == torch.arange(B, device=ids.device).view(1, B)).sum(0).unbind(0))   # no host sync


def _unit(v, fallback, eps):
    """v / (|v| + eps) with the reference's zero-norm fallback rows."""
    norm = torch.norm(v, dim=1)
    if eps:
        norm = norm + eps
    out = torch.div(v, norm.view(-1, 1))
    fb = out.new_zeros((1, 3))
    fb[0, fallback.index(1.0)] = 1.0      # the fallbacks are unit axes: built on the device (no host->device copy)
    return torch.where(torch.eq(norm, 0).view(-1, 1), fb.expand_as(out), out)


def gripper_frame(grasp):
    """grasp (n,>=7) = [centre | axis_y | theta | ...] -> centre (n,3), rotation (n,3,3) whose rows
    are [approach; axis_y; minor_normal] (gripper_region_network.py:447-506)."""
    n = grasp.shape[0]
    center = grasp[:, 0:3].float()
    angle = grasp[:, 6].float()
    cos_t, sin_t = torch.cos(angle).view(n, 1), torch.sin(angle).view(n, 1)
    one, zero = torch.ones_like(cos_t), torch.zeros_like(cos_t)
    R1 = torch.cat((cos_t, zero, -sin_t, zero, one, zero, sin_t, zero, cos_t), dim=1).view(n, 3, 3)
    axis_y = _unit(grasp[:, 3:6].float(), [0.0, 1.0, 0.0], 1e-12)
    axis_x = _unit(torch.cat((axis_y[:, 1:2], -axis_y[:, 0:1], zero), 1), [1.0, 0.0, 0.0], 1e-12)
    axis_z = _unit(torch.cross(axis_x, axis_y, dim=1), [0.0, 0.0, 1.0], 0.0)
    matrix = torch.bmm(torch.stack((axis_x, axis_y, axis_z), dim=2), R1)
    approach = _unit(matrix[:, :, 0], [1.0, 0.0, 0.0], 1e-12)
    minor_normal = torch.cross(approach, axis_y, dim=1)
    return center, torch.stack((approach, axis_y, minor_normal), dim=1)


def _half_extent(value, n, device):
    """Scalar or per-grasp tensor limit -> (n,) float32 of value/2 (:512-516)."""
    if isinstance(value, torch.Tensor):
        return (value.float().view(-1) / 2).to(device).expand(n).contiguous()
    return torch.full((n,), value / 2, dtype=torch.float32, device=device)


def get_gripper_region_transform(group_points, group_index, grasp, region_num, gripper_params, points_too=True):
    """Points of every group that fall inside the predicted gripper's closing box, resampled to
    ``region_num`` per grasp (gripper_region_network.py:436-550).

    group_points (n,G,C), group_index (n,G), grasp (n,>=7) ->
    gripper_pc (n,region_num,C) int64 (the reference's truncating ``torch.full(..., -1)`` quirk),
    gripper_pc_index (n,region_num), gripper_pc_index_inall (n,region_num), valid grasp ids.
    A grasp is valid when more than 5 points are in the box; sampling is without replacement
    when more than ``region_num`` candidates exist, else with replacement.
    ``points_too=False`` (the network's own call, which uses the scene indices only): the first two results are None."""
    widths, height, depths = gripper_params
    n, G, C = group_points.shape
    dev = group_points.device
    native = group_points.is_cuda and grasp.dtype == torch.float32
    center, rot = region_ops.gripper_frame(grasp.to(dev)) if native else gripper_frame(grasp.to(dev))
    xlim, ylim = _half_extent(depths, n, dev), _half_extent(widths, n, dev)
    cand, count = region_ops.box_candidates(group_points, center, rot, xlim, ylim, height / 2)

    # numpy-stream-compatible draws in grasp order, on the device (no synchronisation):
    # > region_num candidates: without replacement; 6..region_num: with replacement; <= 5: invalid
    from . import get_regiondataset as _grd
    valid_ids = None
    if _grd.DEVICE_DRAWS and count.is_cuda:
        pos_t, valid_t = np_random.choice_rows_device(count.int(), region_num, 1, G)
    else:
        np_random.flush()
        if count.is_cuda:
            pos_pinned, valid = np_random.choice_rows_pinned(count.cpu().numpy(), region_num, 1)
            pos_t = pos_pinned.to(dev, non_blocking=True)
            # (the host knows which crops are valid: no device nonzero; flags and ids in one transfer)
            valid_t, valid_ids = host_io.upload_many((valid, np.nonzero(valid)[0]), dev)
        else:
            pos, valid = np_random.choice_rows(count.cpu().numpy(), region_num, 1)
            pos_t, valid_t, valid_ids = host_io.upload_many((pos, valid, np.nonzero(valid)[0]), dev)
    if valid_ids is None:
        valid_ids = torch.nonzero(valid_t).view(-1)     # data-dependent length: the one synchronisation of the crop

    if native and not points_too and group_index.dtype == torch.int64:
        _, index_inall = region_ops.crop_pick(cand, pos_t, valid_t, group_index)
        return None, None, index_inall, valid_ids
    # positions inside the group; rows without a valid crop hold unwritten candidate slots -> 0
    index = torch.where(valid_t.view(n, 1), torch.gather(cand, 1, pos_t).long(), torch.zeros_like(pos_t))
    index_inall = torch.gather(group_index.long(), 1, index)
    picked = torch.gather(group_points, 1, index.unsqueeze(-1).expand(n, region_num, C))
    local = torch.bmm(rot, (picked[:, :, :3].float() - center.view(n, 1, 3)).permute(0, 2, 1)).permute(0, 2, 1)
    gripper_pc = torch.cat((local, picked[:, :, 3:]), -1).to(torch.int64)

    minus1 = torch.full((1,), -1, dtype=torch.int64, device=dev)
    gripper_pc = torch.where(valid_t.view(n, 1, 1), gripper_pc, minus1.view(1, 1, 1))
    index = torch.where(valid_t.view(n, 1), index, minus1.view(1, 1))
    index_inall = torch.where(valid_t.view(n, 1), index_inall, minus1.view(1, 1))
    return gripper_pc, index, index_inall, valid_ids


def _enumerate_templates():
    """(1,4,1,4) fp16 anchor templates: four (+-1/sqrt3) orientations, theta 0
    (gripper_region_network.py:552-587).  The fp16 rounding (0.5771484375) is part of the model."""
    s = math.sqrt(3) / 3
    t_r = torch.tensor([[s, s, s], [s, s, -s], [s, -s, -s], [s, -s, s]], dtype=torch.float32).view(1, 4, 1, 3)
    t_theta = torch.zeros(1, 4, 1, 1, dtype=torch.float32)
    return torch.cat([t_r, t_theta], dim=3).half()


def compute_cos_sim(a, b):
    """1 - cos(a, b) per row, (N,3) x (N,3) -> (N,1) (gripper_region_network.py:589-610)."""
    eps = 1e-12
    dot = torch.sum(a * b, dim=1)
    na = torch.sum(a * a, dim=1) + eps
    nb = torch.sum(b * b, dim=1) + eps
    return (1 - dot / torch.sqrt(na * nb)).view(-1, 1)
