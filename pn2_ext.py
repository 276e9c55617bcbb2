"""Drop-in ``pn2_ext``: the reference does ``import pn2_ext`` at multi_model/utils/pn2_utils/function.py:2.
With this repository on sys.path that import resolves to the MI355X HIP implementation."""
from regnet_for_3d_grasping_amd.pn2_ext import (  # noqa: F401
    ball_query, farthest_point_sample, group_points_backward, group_points_forward, interpolate_backward,
    interpolate_forward, point_search)
