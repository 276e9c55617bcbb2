"""Initialisers (mirror of pn2_utils/nn/init.py:4-46)."""
from torch import nn

_BN_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)


def init_bn(module):
    """BatchNorm affine -> weight 1, bias 0 (init.py:4-8)."""
    if module.weight is not None:
        nn.init.ones_(module.weight)
    if module.bias is not None:
        nn.init.zeros_(module.bias)


def set_bn(module, momentum):
    for m in module.modules():
        if isinstance(m, _BN_TYPES):
            m.momentum = momentum


def _weight_init(fn, **kw):
    def apply(module):
        if module.weight is not None:
            fn(module.weight, **kw)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    return apply


xavier_uniform = _weight_init(nn.init.xavier_uniform_)
xavier_normal = _weight_init(nn.init.xavier_normal_)
kaiming_uniform = _weight_init(nn.init.kaiming_uniform_, nonlinearity="relu")
kaiming_normal = _weight_init(nn.init.kaiming_normal_, nonlinearity="relu")
