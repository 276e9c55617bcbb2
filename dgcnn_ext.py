"""Drop-in ``dgcnn_ext`` (multi_model/utils/pn2_utils/functions/gather_knn.py:2-6)."""
from regnet_for_3d_grasping_amd.dgcnn_ext import gather_knn_backward, gather_knn_forward  # noqa: F401
