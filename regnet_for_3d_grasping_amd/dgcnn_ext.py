"""``dgcnn_ext`` for MI355X (multi_model/utils/pn2_utils/functions/csrc/main.cpp:3-6).

Used only by EdgeFeatureInterpolator, which no REGNet network instantiates; kept for API
completeness on top of the group_points kernels.
"""
import torch

from . import _lib
from .pn2_ext import _eq, _need_f32, _need_i64, _stream

_check = _lib.check
_L = _lib.lib


def gather_knn_forward(input, index):
    """input (B,C,N), index (B,NI,K) -> (B,C,NI,K).  gather_knn_kernel.cu:27-50."""
    _need_f32(input, "input")
    _need_i64(index, "index")
    _eq(input.dim(), 3, "input.dim() does not equal to 3")
    _eq(index.dim(), 3, "index.dim() does not equal to 3")
    _eq(index.size(0), input.size(0), "index.size(0) does not equal to batch_size")
    B, C, N = input.shape
    _, NI, K = index.shape
    with torch.cuda.device(input.device):
        idx = index.contiguous()
        out = torch.empty((B, C, NI, K), dtype=torch.float32, device=input.device)
        _check(_L.regnet_gather_knn_fwd_f32(input.data_ptr(), *input.stride(), idx.data_ptr(), B, C, N, NI, K,
                                            out.data_ptr(), _stream(input)), "gather_knn_forward")
    return out


def gather_knn_backward(grad_output, index):
    """grad_output (B,C,N,K), index (B,N,K) -> (B,C,N).  gather_knn_kernel.cu:100-153."""
    _need_f32(grad_output, "grad_output")
    _need_i64(index, "index")
    _eq(grad_output.dim(), 4, "grad_output.dim() does not equal to 4")
    _eq(index.dim(), 3, "index.dim() does not equal to 3")
    B, C, N, K = grad_output.shape
    _eq(index.size(0), B, "index.size(0) does not equal to batch_size")
    _eq(index.size(2), K, "index.size(2) does not equal to k")
    NI = index.size(1)
    with torch.cuda.device(grad_output.device):
        idx = index.contiguous()
        grad_in = torch.empty((B, C, N), dtype=torch.float32, device=grad_output.device)
        # grad_output rows follow the index rows (NI); the reference sizes grad_input by N
        _check(_L.regnet_gather_knn_bwd_f32(grad_output.data_ptr(), *grad_output.stride(), idx.data_ptr(), B, C, N,
                                            NI, K, grad_in.data_ptr(), _stream(grad_output)), "gather_knn_backward")
    return grad_in
