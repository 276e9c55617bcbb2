"""numpy-compatible batched ``np.random.choice`` draws on numpy's GLOBAL generator, executed by
native host code (csrc/np_random.hip).  Semantics and stream consumption equal the reference's
per-row Python loops; see regnet_np_choice_rows in include/regnet_hip.h."""
import ctypes

import numpy as np

from . import _lib

_check = _lib.check
_L = _lib.lib


def choice_rows(counts, size, mode):
    """counts: int array (any shape) of candidate-list lengths, processed in C order.
    Returns (positions int64 counts.shape + (size,), valid bool counts.shape)."""
    counts32 = np.ascontiguousarray(counts, dtype=np.int32)
    rows = counts32.size
    out = np.empty((rows, size), dtype=np.int64)
    valid = np.empty((rows,), dtype=np.uint8)
    name, key, pos, has_gauss, cached = np.random.get_state()
    if name != "MT19937":
        raise RuntimeError("numpy's global generator is not MT19937")
    key = np.ascontiguousarray(key, dtype=np.uint32).copy()
    cpos = ctypes.c_int32(int(pos))
    _check(_L.regnet_np_choice_rows(key.ctypes.data, ctypes.addressof(cpos), counts32.ctypes.data, rows, int(size),
                                    int(mode), out.ctypes.data, valid.ctypes.data), "np_choice_rows")
    np.random.set_state((name, key, int(cpos.value), has_gauss, cached))
    return out.reshape(counts32.shape + (size,)), valid.astype(bool).reshape(counts32.shape)
