"""Conv/FC + BatchNorm + ReLU blocks and the (Shared)MLP stacks built from them.

Mirror of pn2_utils/nn/modules/{conv.py:6-76, linear.py:6-40, mlp.py:8-114}: same class
names, constructor arguments and sub-module attribute names (``conv``/``fc``, ``bn``, ``relu``),
hence identical ``state_dict`` keys (``mlp.<i>.conv.weight``, ``mlp.<i>.bn.running_mean`` ...).
The affine layer is bias-free whenever BN follows (conv.py:24,64); BN uses eps 1e-5 and the
given momentum; conv weights keep torch's default init unless ``init_weights(fn)`` is called.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .init import init_bn


class _AffineBNReLU(nn.Module):
    """``x -> relu(bn(affine(x)))`` with optional bn / relu; ``_affine_name`` is 'conv' or 'fc'."""

    _affine_name = "conv"

    def _assemble(self, affine, bn, relu):
        setattr(self, self._affine_name, affine)
        self.bn = bn
        self.relu = nn.ReLU(inplace=True) if relu else None

    def forward(self, x, pool_max=False):
        """``pool_max``: also take the max over the last axis (the set-abstraction reduction, modules.py:245)."""
        return self.after_affine(self.affine_only(x), pool_max)

    def affine_only(self, x):
        """The block's convolution / linear layer alone."""
        affine = getattr(self, self._affine_name)
        if self.training and x.is_cuda and self._affine_name == "conv":
            from ... import conv1x1_train
            if conv1x1_train.supported(affine, x):      # (a training BatchNorm follows: its statistics come with the output)
                return conv1x1_train.conv1x1(affine, x, stats=self.bn is not None)
            return affine(x)
        return affine(x)

    def after_affine(self, x, pool_max=False):
        """BatchNorm -> ReLU (-> max over the last axis) of the block, given the affine layer's output.  Callers that
        evaluate the (linear) affine layer before a gather / interpolation enter here."""
        if self.bn is not None and self.training and x.is_cuda:
            # training on the GPU: BatchNorm + ReLU (+ the max over the neighbours) as fused HIP passes
            from ... import bn_train
            group = x.shape[-1] if pool_max and x.dim() == 4 else 0
            if group and bn_train.supported(self.bn, x, group):
                return bn_train.bn_relu(self.bn, x, self.relu is not None, group)
            if bn_train.supported(self.bn, x):
                x = bn_train.bn_relu(self.bn, x, self.relu is not None)
                return torch.max(x, x.dim() - 1)[0] if pool_max else x
        if self.bn is not None:
            x = self.bn(x)
        if self.relu is not None:
            x = self.relu(x)
        return torch.max(x, x.dim() - 1)[0] if pool_max else x

    def init_weights(self, init_fn=None):
        if init_fn is not None:
            init_fn(getattr(self, self._affine_name))
        if self.bn is not None:
            init_bn(self.bn)


class Conv1d(_AffineBNReLU):
    def __init__(self, in_channels, out_channels, kernel_size, relu=True, bn=True, bn_momentum=0.1, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self._assemble(nn.Conv1d(in_channels, out_channels, kernel_size, bias=(not bn), **kwargs),
                       nn.BatchNorm1d(out_channels, momentum=bn_momentum) if bn else None, relu)
        self.init_weights()


class Conv2d(_AffineBNReLU):
    def __init__(self, in_channels, out_channels, kernel_size, relu=True, bn=True, bn_momentum=0.1, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self._assemble(nn.Conv2d(in_channels, out_channels, kernel_size, bias=(not bn), **kwargs),
                       nn.BatchNorm2d(out_channels, momentum=bn_momentum) if bn else None, relu)
        self.init_weights()


class FC(_AffineBNReLU):
    _affine_name = "fc"

    def __init__(self, in_channels, out_channels, relu=True, bn=True, bn_momentum=0.1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self._assemble(nn.Linear(in_channels, out_channels, bias=(not bn)),
                       nn.BatchNorm1d(out_channels, momentum=bn_momentum) if bn else None, relu)
        # note: the reference FC does not call init_weights() in its constructor (linear.py:18-28)


class _Stack(nn.ModuleList):
    """A ModuleList of blocks applied in order, with functional dropout after every block in
    training mode only (mlp.py:41-47, :95-107)."""

    def _dropout(self, x):
        return F.dropout(x, p=self.dropout_prob, training=True)

    def forward(self, x, pool_max=False, first_affine_done=False):
        """``pool_max``: return the max over the last axis of the stack's output (fused into the last block's
        BatchNorm + ReLU pass when training on the GPU).  ``first_affine_done``: ``x`` already is the output of the
        first block's convolution (evaluated by the caller before a gather / interpolation, which it commutes with)."""
        last = len(self) - 1
        dropout = self.training and self.dropout_prob > 0.0
        pending = None          # (bn_train.Pending, activation shape): the previous block's BatchNorm + ReLU, not applied yet
        for i, block in enumerate(self):
            fuse = pool_max and i == last and not dropout
            if pending is not None:
                from ... import conv1x1_train
                x = conv1x1_train.conv1x1_of_pending(block.conv, pending[0], pending[1], stats=block.bn is not None)
                pending = None
            elif not (i == 0 and first_affine_done):
                x = block.affine_only(x)
            # x is the block's convolution output.  conv -> bn -> relu -> conv without the normalised activation in between
            # (training on the GPU): this block only takes the batch statistics, the next block's convolution applies them
            if i < last and not dropout and self._defers(block, self[i + 1], x):
                from ... import bn_train
                pending = (bn_train.bn_stats(block.bn, x, block.relu is not None), x.shape)
                continue
            x = block.after_affine(x, pool_max=fuse)
            if dropout:
                x = self._dropout(x)
        if pool_max and (dropout or last < 0):
            x = torch.max(x, x.dim() - 1)[0]
        return x

    @staticmethod
    def _defers(block, nxt, x):
        if not (block.training and x.is_cuda and x.dtype == torch.float32 and block.bn is not None
                and block._affine_name == "conv" and nxt._affine_name == "conv"):
            return False
        from ... import bn_train, conv1x1_train
        return bn_train.supported(block.bn, x) and conv1x1_train.pending_ok(nxt.conv, x)

    def init_weights(self, init_fn=None):
        for block in self:
            block.init_weights(init_fn)

    def extra_repr(self):
        return "dropout_prob={}".format(self.dropout_prob) if self.dropout_prob > 0.0 else ""


class MLP(_Stack):
    def __init__(self, in_channels, mlp_channels, dropout_prob=0.0, bn=True, bn_momentum=0.1):
        super().__init__()
        assert dropout_prob >= 0.0
        self.in_channels, self.out_channels, self.dropout_prob = in_channels, mlp_channels[-1], dropout_prob
        for width in mlp_channels:
            self.append(FC(in_channels, width, relu=True, bn=bn, bn_momentum=bn_momentum))
            in_channels = width


class SharedMLP(_Stack):
    """1x1-conv MLP shared over one (``ndim=1``: (B,C,N)) or two (``ndim=2``: (B,C,N,K)) point axes."""

    def __init__(self, in_channels, mlp_channels, ndim=1, dropout_prob=0.0, bn=True, bn_momentum=0.1):
        super().__init__()
        if ndim not in (1, 2):
            raise ValueError("SharedMLP only supports ndim=(1, 2).")
        assert dropout_prob >= 0.0
        self.in_channels, self.out_channels = in_channels, mlp_channels[-1]
        self.ndim, self.dropout_prob = ndim, dropout_prob
        block = Conv1d if ndim == 1 else Conv2d
        for width in mlp_channels:
            self.append(block(in_channels, width, 1, relu=True, bn=bn, bn_momentum=bn_momentum))
            in_channels = width

    def _dropout(self, x):
        if self.ndim == 1:
            return F.dropout(x, p=self.dropout_prob, training=True)
        return F.dropout2d(x, p=self.dropout_prob, training=True)
