"""Import-path alias: the reference's module path, served by regnet_for_3d_grasping_amd."""
from regnet_for_3d_grasping_amd.pn2_utils.function import *  # noqa: F401,F403
from regnet_for_3d_grasping_amd.pn2_utils.function import (  # noqa: F401
    BallQuery, FarthestPointSample, FeatureInterpolate, GroupPoints, SearchNNDistance, ball_query,
    farthest_point_sample, feature_interpolate, gather_points, group_points, search_nn_distance)
