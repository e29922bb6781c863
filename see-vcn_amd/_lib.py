"""ctypes binding of libseevcn_hip.so (the C-ABI in include/seevcn_hip.h).

There is no CPU fallback: if the library is missing or a symbol is absent, import of any op fails loudly.
Tensors are passed as raw device pointers plus the current torch HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEEVCN_LIB") or os.path.join(_HERE, "lib", "libseevcn_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "seevcn_hip.h")

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_f = ctypes.c_float
c_d = ctypes.c_double
c_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/seevcn_hip.h declares
SIGNATURES = {
    "sv_abi_version": (c_i, []),
    "sv_last_error": (ctypes.c_char_p, []),
    "sv_index_persistent_bytes": (c_sz, [c_i64]),
    "sv_index_scratch_bytes": (c_sz, [c_i64]),
    "sv_voxelize_dynamic_scratch_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "sv_voxelize_dynamic": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p]),
    "sv_mean_vfe": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p, c_p]),
    "sv_fill_f32": (c_i, [c_p, c_i64, c_f, c_p]),
    "sv_mean_square_scratch_bytes": (ctypes.c_size_t, []),
    "sv_mean_square": (c_i, [c_p, c_i64, c_f, c_f, c_p, c_p, c_p, c_p]),
    "sv_scale_by_device_scalar": (c_i, [c_p, c_i64, c_p, c_p]),
    "sv_gemm_bias_act": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "sv_gemm_splitk_splits": (c_i, [c_i, c_i, c_i]),
    "sv_gemm_splitk_scratch_bytes": (c_sz, [c_i, c_i, c_i]),
    "sv_gemm_bias_act_splitk": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p]),
    "sv_pointwise_conv3": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_p]),
    "sv_pointwise_conv3_gather": (c_i, [c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p]),
    "sv_gemm_tn_scratch_bytes": (c_sz, [c_i64, c_i, c_i]),
    "sv_gemm_tn": (c_i, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i64, c_i, c_i, c_p, c_p]),
    "sv_gemm_strided_scratch_bytes": (c_sz, [c_i, c_i, c_i64]),
    "sv_gemm_strided": (c_i, [c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_i, c_i, c_i64, c_p, c_p]),
    "sv_column_sums_scratch_bytes": (c_sz, [c_i64, c_i]),
    "sv_column_sums": (c_i, [c_p, c_i64, c_i64, c_i, c_p, c_p, c_p]),
    "sv_segment_max": (c_i, [c_p, c_i64, c_i, c_i, c_i, c_p, c_p, c_p]),
    "sv_segment_max_backward": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_i64, c_p]),
    "sv_segment_sum": (c_i, [c_p, c_i64, c_i, c_i, c_i, c_p, c_p]),
    "sv_act_backward": (c_i, [c_p, c_p, c_i64, c_i, c_f, c_p, c_p]),
    "sv_vcn_vc_prep": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_vcn_vc_pose": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_vcn_vc_finish": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sv_vcn_cn_transform": (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_p]),
    "sv_conv_out_shape": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p]),
    "sv_rulebook_scratch_bytes": (c_sz, [c_i64, c_i64]),
    "sv_rulebook_subm": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "sv_cellmap_persistent_bytes": (c_sz, [c_i, c_p]),
    "sv_rulebook_subm_cellmap": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "sv_rulebook_sparse": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p]),
    "sv_rulebook_chain_scratch_bytes": (c_sz, [c_i64]),
    "sv_rulebook_chain_count": (c_i, [c_p, c_i64, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "sv_rulebook_batch": (c_i, [c_p, c_i, c_p]),
    "sv_rulebook_invert": (c_i, [c_p, c_i64, c_i, c_p, c_i64, c_p]),
    "sv_rulebook_invert_rows": (c_i, [c_p, c_i64, c_i, c_p, c_i64, c_p]),
    "sv_rulebook_pair_counts": (c_i, [c_p, c_i64, c_i, c_p, c_p]),
    "sv_sparse_conv_gather_gemm": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p]),
    "sv_conv_plan_persistent_bytes": (c_sz, []),
    "sv_conv_table_rows": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p]),
    "sv_conv_plan_perm_bytes": (c_sz, [c_i64]),
    "sv_conv_plan_build": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p]),
    "sv_sparse_conv_gather_gemm_planned": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i64, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    "sv_sparse_conv_dgrad_planned_bn": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    "sv_conv_planned_partials": (c_i, []),
    "sv_batchnorm_relu_forward_partial": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sv_debug_conv_trace": (c_i, [c_p]),
    "sv_debug_wgrad_trace": (c_i, [c_p]),
    "sv_conv_tiles_per_wave": (c_i, [c_i64, c_i, c_i]),
    "sv_conv_plan_tiles_bytes": (c_sz, [c_i64, c_i]),
    "sv_conv_weight_fragments_batch": (c_i, [c_p, c_i, c_i64, c_p]),
    "sv_conv_plan_build_dealt": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_p]),
    "sv_conv_plan_tiles": (c_i, [c_p, c_i64, c_i, c_p, c_p]),
    "sv_conv_plan_build_dealt_batch": (c_i, [c_p, c_i, c_p]),
    "sv_conv_mfma_kernel_applies": (c_i, [c_i, c_i, c_i, c_i64]),
    "sv_conv_weight_fragments": (c_i, [c_p, c_i64, c_i64, c_i64, c_i, c_i, c_i, c_p, c_p, c_p]),
    "sv_sparse_conv_wgrad_scratch_bytes": (c_sz, [c_i64, c_i, c_i, c_i]),
    "sv_sparse_conv_wgrad": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p, c_p]),
    "sv_sparse_conv_wgrad_strided": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_p, c_p]),
    "sv_sparse_conv_wgrad_partial_bytes": (c_sz, [c_i64, c_i, c_i, c_i]),
    "sv_sparse_conv_wgrad_stage1": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "sv_sparse_conv_wgrad_reduce_batch": (c_i, [c_p, c_i, c_p]),
    "sv_wgrad_plan_bytes": (c_sz, [c_i64, c_i, c_i]),
    "sv_wgrad_plan_pieces": (c_i, [c_i, c_i]),
    "sv_wgrad_plan_build": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p]),
    "sv_wgrad_plan_build_batch": (c_i, [c_p, c_i, c_p]),
    "sv_wgrad_planned_applies": (c_i, [c_i64, c_i64, c_i, c_i, c_i]),
    "sv_sparse_conv_wgrad_planned_bytes": (c_sz, [c_i, c_i, c_i]),
    "sv_sparse_conv_wgrad_planned": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "sv_sparse_conv_wgrad_planned_stage1": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p]),
    "sv_sparse_to_dense_scratch_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "sv_sparse_to_dense": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "sv_dense_to_sparse": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "sv_sparse_to_dense_nhwc_applies": (c_i, [c_i, c_i, c_i, c_i]),
    "sv_sparse_to_dense_nhwc": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "sv_dense_to_sparse_nhwc": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "sv_farthest_point_sampling": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "sv_stack_farthest_point_sampling": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "sv_fps_multi_scratch_bytes": (c_sz, [c_i]),
    "sv_stack_farthest_point_sampling_multi": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_fps_multi_error_offset": (c_sz, [c_i]),
    "sv_stack_farthest_point_sampling_multi_async": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p]),
    "sv_fps_bucket_applies": (c_i, [c_i, c_i, c_i]),
    "sv_fps_bucket_scratch_bytes": (c_sz, [c_i, c_i]),
    "sv_farthest_point_sampling_bucketed": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "sv_ball_query_stack": (c_i, [c_i, c_i, c_i, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "sv_ball_query_hash_scratch_bytes": (c_sz, [c_i64]),
    "sv_ball_query_stack_hashed": (c_i, [c_i, c_i, c_i64, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "sv_sa_prepare_weights": (c_i, [c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_p, c_p, c_p]),
    "sv_sa_mlp_max": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p]),
    "sv_sa_train_scratch_bytes": (c_sz, [c_i, c_i, c_i]),
    "sv_sa_train_forward": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f,
                                  c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "sv_sa_train_backward": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p,
                                   c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "sv_group_points_stack": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sv_group_points_grad_stack": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sv_boxes_overlap_bev": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p]),
    "sv_boxes_iou3d_batch": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_p, c_p]),
    "sv_nms_scratch_bytes": (c_sz, [c_i]),
    "sv_nms": (c_i, [c_p, c_i, c_f, c_i, c_p, c_p, c_p, c_p]),
    "sv_nms_prefix": (c_i, [c_p, c_i, c_f, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_points_in_boxes": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "sv_voxelize_hard_scratch_bytes": (c_sz, [c_i, c_i64, c_i]),
    "sv_voxelize_hard": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i64, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "sv_pillar_decorate": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p]),
    "sv_bev_interpolate": (c_i, [c_p, c_i64, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_p, c_p]),
    "sv_bev_interpolate_grad_scratch_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "sv_sigmoid_focal_loss": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_f, c_f, c_p, c_p, c_p]),
    "sv_weighted_smooth_l1_loss": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_f, c_p, c_p, c_p]),
    "sv_bev_interpolate_grad": (c_i, [c_p, c_i64, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_p]),
    "sv_center_assign_targets": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_i, c_p, c_p, c_p,
                                       c_p, c_p]),
    "sv_vcn_surface_select_scratch_bytes": (c_sz, [c_i]),
    "sv_vcn_surface_select": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_vcn_largest_cluster": (c_i, [c_p, c_i, c_i, c_d, c_i, c_i, c_p, c_p, c_p]),
    "sv_vcn_largest_cluster_periodic": (c_i, [c_p, c_i, c_i, c_p, c_d, c_i, c_i, c_p, c_p, c_p]),
    "sv_dedup_rows_scratch_bytes": (c_sz, [c_i64]),
    "sv_dedup_rows": (c_i, [c_p, c_i64, c_p, c_p]),
    "sv_points_near_set": (c_i, [c_p, c_i64, c_p, c_i64, c_i, c_d, c_p, c_p]),
    "sv_points_near_set_scratch_bytes": (c_sz, [c_i64]),
    "sv_points_near_set_boxed": (c_i, [c_p, c_i64, c_p, c_i64, c_i, c_d, c_p, c_p, c_p]),
    "sv_points_in_boxes_matrix": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p]),
    "sv_crop_points_in_boxes": (c_i, [c_p, c_i64, c_i, c_p, c_i, c_i64, c_p, c_p, c_p]),
    "sv_project_lidar_to_image_kitti": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_i, c_i, c_d, c_p, c_p, c_p, c_p]),
    "sv_project_lidar_to_image_camera": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_project_lidar_to_image_nuscenes": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_i, c_i, c_d, c_p, c_p, c_p, c_p]),
    "sv_polygon_masks_scratch_bytes": (c_sz, [c_i, c_i, c_i]),
    "sv_polygons_to_masks": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "sv_polygons_to_masks_shrunk": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_points_in_masks": (c_i, [c_p, c_p, c_i64, c_p, c_p, c_i, c_i, c_i, c_i64, c_p, c_p, c_p]),
    "sv_isolate_cluster_scratch_bytes": (c_i64, [c_i, c_i64]),
    "sv_isolate_largest_cluster": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_i64, c_d, c_d, c_d, c_d, c_d, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sv_gemm_bias_act_ragged": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "sv_gemm_bias_act_ragged_dev": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_f, c_p]),
    "sv_unique_rows": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p]),
    "sv_unique_rows_compact": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_ball_query_batch": (c_i, [c_i, c_i, c_i, c_f, c_i, c_p, c_p, c_p, c_p]),
    "sv_group_points_batch": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_group_points_grad_batch": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_gather_points_batch": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_gather_points_grad_batch": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "sv_three_nn_batch": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sv_three_interpolate_batch": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sv_three_interpolate_grad_batch": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sv_chamfer_forward": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sv_chamfer_backward": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "sv_batchnorm_scratch_bytes": (c_sz, [c_i]),
    "sv_batchnorm_relu_forward": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "sv_batchnorm_relu_backward": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sv_batchnorm_relu_backward_partial": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p]),
    "sv_conv_next_input_norm": (c_i, [c_p, c_i]),
    "sv_batchnorm_finalize_forward": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sv_batchnorm_apply": (c_i, [c_p, c_i64, c_i, c_p, c_i, c_p, c_p]),
    "sv_run_ops": (c_i, [c_p, c_i, c_p]),
    "sv_run_ops_timed": (c_i, [c_p, c_i, c_p, c_p]),
    "sv_run_ops_two_streams": (c_i, [c_p, c_i, c_p, c_p]),
    "sv_anchor_decode": (c_i, [c_p, c_i64, c_p, c_p, c_i, c_i, c_f, c_f, c_p, c_p]),
    "sv_assign_targets_axis_aligned": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
}

_lib = None


class SeevcnHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and attach signatures. Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SeevcnHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C see-vcn_amd/csrc`). seevcn_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().sv_last_error().decode(errors="replace")
        raise SeevcnHipError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "seevcn_amd ops need contiguous tensors"
    return t.data_ptr()


def stream():
    """Raw handle of the calling thread's current stream.  (torch.cuda.current_stream() builds a Stream object through four layers of device
    index helpers: 10 us a call, 74 calls per bench step = 0.7 ms of a host thread that the step is bound by.)"""
    return _raw_stream(_get_device())


_raw_stream = torch._C._cuda_getCurrentRawStream
_get_device = torch._C._cuda_getDevice


import threading as _threading

_sync_hooks = _threading.local()


def _take_checks():
    """The parked checks whose value the CURRENT stream orders (parked on it); checks parked on another stream stay parked -- read here they
    would race with the stream that still computes them (bench.py runs a piece of the trained side, main stream, inside a read of the
    input side, side stream: a lazily built rulebook there must not pick up the input side's pending empty-cluster check)."""
    pending = getattr(_sync_hooks, "checks", None)
    if not pending:
        return []
    cur = stream()
    mine = [c for c in pending if c[0] == cur]
    if mine:
        _sync_hooks.checks = [c for c in pending if c[0] != cur]
    return mine


def host_int(t):
    """int(t.item()) -- a device -> host read that blocks the calling thread until the stream has produced t.  A caller that has other work to
    enqueue meanwhile (bench.py: the trained side of the previous batch) installs a hook with set_sync_hook(); it runs right before the read,
    i.e. after the kernels that produce t were launched, so the wait is spent enqueueing instead of idling.  Checks parked with defer_check()
    on the same stream ride on this read: their values come back in the same copy."""
    return host_ints([t])[0]


def host_ints(tensors):
    """The integers of `tensors` (0-dim / one-element tensors, or longer ones: their elements in order) as one flat Python list with ONE blocking
    device -> host read: one concatenation kernel + one copy (same hook and parked checks as host_int)."""
    hook = getattr(_sync_hooks, "before", None)
    if hook is not None:
        hook()
    pending = _take_checks()
    if not pending and len(tensors) == 1 and tensors[0].numel() == 1:
        return [int(tensors[0].item())]
    flat = [t.reshape(-1) for t in tensors] + [p.reshape(-1) for _, p, _ in pending]
    if len({t.dtype for t in flat}) > 1:
        flat = [t.to(torch.int64) for t in flat]
    vals = torch.cat(flat).tolist()
    n = len(vals) - len(pending)
    for v, (_, _, fn) in zip(vals[n:], pending):
        fn(int(v))
    return [int(v) for v in vals[:n]]


def defer_check(t, fn):
    """Park a validity check on a 0-dim device integer: fn(value) runs (and may raise) at the calling thread's next host_int() or
    flush_checks() ON THE SAME STREAM -- whose order guarantees t is final by then -- instead of costing a blocking read of its own."""
    if getattr(_sync_hooks, "checks", None) is None:
        _sync_hooks.checks = []
    _sync_hooks.checks.append((stream(), t, fn))


def flush_checks():
    """Run the checks parked on the current stream now (one blocking read) -- for call sites that are not followed by a host_int()."""
    pending = _take_checks()
    if pending:
        flat = [p.reshape(-1) for _, p, _ in pending]
        if len({t.dtype for t in flat}) > 1:
            flat = [t.to(torch.int64) for t in flat]
        for v, (_, _, fn) in zip(torch.cat(flat).tolist(), pending):
            fn(int(v))


def set_sync_hook(fn):
    """Install (or with None remove) the calling thread's before-read hook; returns the previous one."""
    prev = getattr(_sync_hooks, "before", None)
    _sync_hooks.before = fn
    return prev


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SeevcnHipError("seevcn_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def host_array(ctype, values):
    return (ctype * len(values))(*values)


class Workspace:
    """Grow-only device byte buffers, one set per (device, stream).

    `persistent(name, nbytes)` returns a zero-initialised buffer that the kernels keep zeroed between calls
    (coordinate-index bitmaps); `scratch(name, nbytes)` returns uninitialised bytes.  Every buffer is used inside ONE op call on the
    calling thread's current stream; the key holds that stream's raw handle, so two streams (bench.py runs the input side of batch N + 1
    beside the trained side of batch N) never share a buffer, and a grow-realloc frees a buffer only on the stream that used it.
    """

    def __init__(self):
        self._bufs = {}

    def _get(self, kind, name, nbytes, device, zero):
        key = (kind, name, device.index, _raw_stream(device.index if device.index is not None else _get_device()))
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            n = int(nbytes * 1.25) + 256 if buf is not None else int(nbytes)
            n = max(n, 256)
            buf = (torch.zeros if zero else torch.empty)(n, dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf

    def persistent(self, name, nbytes, device):
        return self._get("p", name, nbytes, device, True)

    def scratch(self, name, nbytes, device):
        return self._get("s", name, nbytes, device, False)

    def reset(self):
        self._bufs.clear()


workspace = Workspace()
