"""Weight initialisers of the shared-MLP blocks.

API-compatible with pn2_utils/nn/init.py:4-46 of the reference: ``init_bn`` and ``set_bn`` operate
on BatchNorm layers, the four ``<scheme>_<distribution>`` callables take a conv / linear module,
re-draw its weight and zero its bias.  The callables are generated from one table.
"""
from torch import nn
from torch.nn import init as _torch_init

_BN_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)

# public name -> (torch initialiser, keyword arguments)
_SCHEMES = {
    "xavier_uniform": (_torch_init.xavier_uniform_, {}),
    "xavier_normal": (_torch_init.xavier_normal_, {}),
    "kaiming_uniform": (_torch_init.kaiming_uniform_, {"nonlinearity": "relu"}),
    "kaiming_normal": (_torch_init.kaiming_normal_, {"nonlinearity": "relu"}),
}


def init_bn(module):
    """Identity affine for a BatchNorm layer: gamma = 1, beta = 0 (either may be absent)."""
    gamma, beta = module.weight, module.bias
    if gamma is not None:
        _torch_init.ones_(gamma)
    if beta is not None:
        _torch_init.zeros_(beta)


def set_bn(module, momentum):
    """Set the running-statistics momentum of every BatchNorm layer below ``module``."""
    for layer in module.modules():
        if isinstance(layer, _BN_TYPES):
            layer.momentum = momentum


def _make(name):
    draw, kwargs = _SCHEMES[name]

    def initialise(module):
        if module.weight is not None:
            draw(module.weight, **kwargs)
        if module.bias is not None:
            _torch_init.zeros_(module.bias)

    initialise.__name__ = name
    initialise.__doc__ = "Re-draw ``module.weight`` with torch's %s_ and zero ``module.bias``." % name
    return initialise


globals().update({_name: _make(_name) for _name in _SCHEMES})
__all__ = ["init_bn", "set_bn"] + sorted(_SCHEMES)
