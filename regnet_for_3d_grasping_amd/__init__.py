"""MI355X-native PointNet++ set-abstraction + region-grouping hot path for REGNet.

Product code only: hand-written gfx950 HIP kernels (``csrc/``) behind a C ABI
(``include/regnet_hip.h``), a ctypes binding exposing the reference's ``pn2_ext`` /
``dgcnn_ext`` surface (``pn2_ext.py``, ``dgcnn_ext.py``) and the host-side mirror of the
reference's Python operator/model API (``pn2_utils/``, ``pointnet2.py``, ``score_network.py``,
``gripper_region_network.py``, ``get_regiondataset.py``).  There is no CPU fallback: without the
built library / a GPU the ops raise.
"""
__version__ = "0.1.0"
