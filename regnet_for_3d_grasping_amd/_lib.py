"""ctypes loader for ``csrc/libregnet_hip.so`` (the C ABI declared in include/regnet_hip.h).

Fails loudly: a missing library raises ImportError at import time and a non-zero status from
any entry point raises RuntimeError (the reference's TORCH_CHECK / THCudaCheck convention,
e.g. csrc/sampling_kernel.cu:134-137,167).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("REGNET_HIP_LIB") or os.path.join(_HERE, "csrc", "libregnet_hip.so")   # override: A/B builds only

_i64, _f32, _f64, _vp, _int = ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p, ctypes.c_int

# name -> (restype, argtypes); mirrors include/regnet_hip.h one to one.
SIGNATURES = {
    "regnet_abi_version": (_int, []),
    "regnet_build_info": (ctypes.c_char_p, []),
    "regnet_strerror": (ctypes.c_char_p, [_int]),
    "regnet_fps_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "regnet_fps_chain_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "regnet_fps_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "regnet_fps_status_offset_bytes": (_i64, [_i64, _i64, _i64]),
    "regnet_ball_query_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _i64,
                                     _vp, _vp, _vp]),
    "regnet_grid_workspace_bytes": (_i64, [_i64, _i64]),
    "regnet_three_nn_grid_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp,
                                        _vp]),
    "regnet_ball_query_grid_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _i64,
                                          _vp, _vp, _vp, _vp]),
    "regnet_group_points_fwd_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "regnet_group_points_bwd_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp,
                                           _vp]),
    "regnet_three_nn_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "regnet_interpolate_fwd_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "regnet_interpolate_bwd_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "regnet_gather_knn_fwd_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "regnet_gather_knn_bwd_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "regnet_radius_group_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _f32, _i64, _vp, _vp,
                                       _vp]),
    "regnet_select_positive_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _vp]),
    "regnet_box_crop_f32": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _f32, _i64, _i64, _vp, _vp, _vp]),
    "regnet_gather_max_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp]),
    "regnet_gather_max_scene_f32": (_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "regnet_gripper_frame_f32": (_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "regnet_stage2_loss_rows_f32": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _f32, _vp, _vp, _i64, _vp, _vp, _vp,
                                           _vp, _vp, _vp, _vp]),
    "regnet_label_match_f32": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _f32, _f64, _vp, _vp, _vp]),
    "regnet_ce_rows_f32": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp]),
    "regnet_refine_loss_rows_f32": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "regnet_heads_chain_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "regnet_heads_tree_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "regnet_head_layer_train_supported": (_int, [_i64, _i64, _i64]),
    "regnet_head_layer_train_fwd_f32": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i64, _i64, _i64, _int,
                                               _vp, _vp, _vp, _vp]),
    "regnet_head_layer_train_bwd_f32": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _int, _vp, _vp,
                                               _vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "regnet_stage2_decode_f32": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _f32, _int, _i64, _vp, _vp]),
    "regnet_refine_decode_f32": (_int, [_vp, _i64, _vp, _vp, _i64, _f32, _f32, _i64, _vp, _vp, _vp]),
    "regnet_crop_pick": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "regnet_gather_max_arg_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp]),
    "regnet_scatter_max_grad_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "regnet_rowsum_neg_f32": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "regnet_mlp_layer_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _int, _int, _vp]),
    "regnet_mlp_splitk_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "regnet_mlp_layer_splitk_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _int, _i64, _vp,
                                           _vp]),
    "regnet_sa_layer1_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64,
                                    _vp, _i64, _vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "regnet_sa_chain3_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp,
                                    _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _int, _vp, _i64,
                                    _vp]),
    "regnet_grasp_collision_counts_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _f32, _f32, _vp, _f32, _f32, _f32, _f32,
                                                 _vp, _vp]),
    "regnet_grasp_antipodal_stats_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _f32, _f32, _vp, _f32,
                                                _f32, _f32, _f32, _f32, _vp, _vp, _vp]),
    "regnet_resample_groups_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "regnet_normals_workspace_bytes": (_i64, [_i64]),
    "regnet_estimate_normals_f32": (_int, [_vp, _i64, _f64, _i64, _f64, _f64, _f64, _vp, _vp, _vp, _vp]),
    "regnet_bn_workspace_bytes": (_i64, [_i64]),
    "regnet_bn_relu_train_fwd_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _f32, _f32, _vp, _vp, _int, _i64, _vp, _vp, _vp,
                                            _vp, _vp, _vp]),
    "regnet_bn_relu_train_bwd_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _int, _i64, _vp, _vp,
                                            _vp, _vp, _vp]),
    "regnet_bn_train_stats_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "regnet_bn_relu_train_fwd_from_sums_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _f32, _f32, _vp, _vp, _int, _i64, _vp, _vp,
                                                      _vp, _vp, _vp, _vp]),
    "regnet_bn_train_stats_from_sums_f32": (_int, [_i64, _i64, _i64, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "regnet_pack_rows_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "regnet_pack_rows_centred_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp]),
    "regnet_gather_points_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp]),
    "regnet_class_order_i64": (_int, [_vp, _i64, _vp, _vp]),
    "regnet_sa_premul_layer_f32": (_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp,
                                          _vp, _i64, _i64, _int, _int, _vp]),
    "regnet_sa_layer12_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64,
                                     _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _int, _int, _vp]),
    "regnet_interp_concat_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _f32, _vp, _i64, _i64, _i64, _i64, _i64,
                                        _i64, _vp, _i64, _i64, _vp]),
    "regnet_interp_affine_f32": (_int, [_vp, _i64, _i64, _vp, _vp, _f32, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp,
                                        _vp, _vp, _int, _i64, _i64, _i64, _vp, _i64, _vp]),
    "regnet_score_head_f32": (_int, [_vp, _i64, _i64, _vp, _f32, _f32, _f32, _vp, _i64, _vp]),
    "regnet_sa_premul_chain_stream_floats": (_i64, []),
    "regnet_sa_premul_chain_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _int, _vp, _i64,
                                          _vp, _vp]),
    "regnet_sa3_premul_chain_stream_floats": (_i64, []),
    "regnet_sa3_premul_chain_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _int, _vp, _i64,
                                           _vp, _vp]),
    "regnet_fp_head_chain_stream_floats": (_i64, []),
    "regnet_fp_head_chain_blocks": (_i64, [_i64]),
    "regnet_fp_head_chain_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _f32, _f32, _f32, _vp, _i64, _vp, _i64, _vp,
                                        _i64, _i64, _vp]),
    "regnet_fp_head_chain_interp_f32": (_int, [_vp, _i64, _i64, _vp, _vp, _f32, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64,
                                               _vp, _i64, _vp, _i64, _vp, _f32, _f32, _f32, _vp, _i64, _vp, _vp, _i64, _i64,
                                               _vp]),
    "regnet_conv1x1_train_supported": (_int, [_i64, _i64, _i64]),
    "regnet_conv1x1_fwd_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "regnet_conv1x1_dgrad_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "regnet_conv1x1_fwd_stream_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "regnet_conv1x1_dgrad_stream_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "regnet_conv1x1_fwd_smallci_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "regnet_conv1x1_smallci_stats_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "regnet_conv1x1_fwd_smallci_stats_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "regnet_conv1x1_wgrad_smallci_partials": (_i64, [_i64, _i64, _i64]),
    "regnet_conv1x1_wgrad_smallci_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "regnet_conv1x1_smallco_f32": (_int, [_int, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "regnet_conv1x1_stream_reserve_slots": (_int, [_int]),
    "regnet_sa_chain3_split_plane_bytes": (_i64, [_i64]),
    "regnet_sa_chain3_split_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp,
                                         _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _int, _vp, _int, _vp, _i64, _vp, _vp]),
    "regnet_conv1x1_split_supported": (_int, [_i64, _i64, _i64]),
    "regnet_conv1x1_split_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "regnet_conv1x1_split_f32": (_int, [_int, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp]),
    "regnet_conv1x1_bnrelu_supported": (_int, [_i64, _i64, _i64]),
    "regnet_conv1x1_fwd_bnrelu_stream_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp]),
    "regnet_conv1x1_fwd_stats_supported": (_int, [_i64, _i64, _i64, _int]),
    "regnet_conv1x1_fwd_stats_stream_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp]),
    "regnet_conv1x1_wgrad_bnrelu_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp]),
    "regnet_conv1x1_wgrad_slices": (_i64, [_i64, _i64, _i64, _i64]),
    "regnet_conv1x1_wgrad_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "regnet_conv1x1_wgrad_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "regnet_np_choice_rows": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp, _vp]),
    "regnet_np_choice_rows_dev_workspace_ints": (_i64, [_i64, _i64]),
    "regnet_np_choice_rows_dev": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp]),
    "regnet_np_rand_doubles_dev": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "regnet_dataset_resample_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libregnet_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python regnet_for_3d_grasping_amd/csrc/build.py`; there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(status, what):
    """Raise RuntimeError for a non-zero status code of entry point ``what``."""
    if status != 0:
        msg = lib.regnet_strerror(int(status)).decode()
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg, status))
