"""Fused eval-mode forward for SA / FP modules on MI355X (placeholder until mlp.hip lands).

``usable(module, xyz)`` is the single dispatch predicate used by ``pn2_utils.modules``: the
fused path is taken only for inference (module in eval mode, autograd off) on GPU tensors.
"""
import torch

ENABLED = False


def usable(module, xyz):
    return ENABLED and xyz.is_cuda and not module.training and not torch.is_grad_enabled()


def sa_forward(module, xyz, feature):
    raise NotImplementedError


def fp_forward(module, dense_xyz, sparse_xyz, dense_feature, sparse_feature):
    raise NotImplementedError
