"""``gather_knn`` (mirror of pn2_utils/functions/gather_knn.py:9-23) on the MI355X ``dgcnn_ext``."""
from torch.autograd import Function

from ... import dgcnn_ext


class GatherKNN(Function):
    @staticmethod
    def forward(ctx, feature, index):
        ctx.save_for_backward(index)
        return dgcnn_ext.gather_knn_forward(feature, index)

    @staticmethod
    def backward(ctx, grad_output):
        (index,) = ctx.saved_tensors
        return dgcnn_ext.gather_knn_backward(grad_output, index), None


gather_knn = GatherKNN.apply
