/* seevcn_hip.h — C ABI of libseevcn_hip.so (hand-written HIP kernels for gfx950 / MI355X).
 *
 * This is the drop-in boundary for the SEE-VCN hot path: every entry point below is what the
 * reference's Python would bind (ctypes / pybind) in place of the CUDA extension or third-party
 * call named in its comment (file:line relative to the reference tree).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - every entry takes the HIP stream it must enqueue on (`void* stream` = hipStream_t; NULL = null stream)
 *     — the reference launches on the legacy default stream (e.g. pointnet2_stack/src/ball_query_gpu.cu:83);
 *   - entries never allocate, never synchronise and never exit(): they return 0 or an SV_ERR_* code and
 *     sv_last_error() gives the message (the reference prints and calls exit(-1), ball_query_gpu.cu:85-89);
 *   - counts that are only known on the device are written to caller-provided device int32 slots; callers
 *     pass capacities. Scratch/workspace sizes come from the matching *_bytes() query;
 *   - "persistent" workspaces must be zero-filled once by the caller (hipMemset) and are returned zeroed
 *     by every entry that uses them (entries clean exactly the cells they touched).
 *   - float = IEEE fp32, indices = int32, linear cell keys = int64.
 */
#ifndef SEEVCN_HIP_H
#define SEEVCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SV_OK 0
#define SV_ERR_ARG 1
#define SV_ERR_HIP 2

#define SV_ABI_VERSION 1

int sv_abi_version(void);
const char* sv_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Coordinate index (persistent workspace shared by voxelisation and rulebook builds)
 * ---------------------------------------------------------------------------------------------- */
/* bytes of persistent (zero-initialised) workspace indexing `ncells` grid cells.  A workspace belongs to ONE cell count: its layout
 * (occupancy words | chunk counts | chunk bases) depends on ncells and a call returns only the words and counts to zero, so reusing
 * it with another ncells needs a memset in between. */
size_t sv_index_persistent_bytes(int64_t ncells);
/* bytes of per-call scratch used by the chunk scan for `ncells` cells */
size_t sv_index_scratch_bytes(int64_t ncells);

/* ------------------------------------------------------------------------------------------------
 * Dynamic voxelisation + mean VFE
 *   replaces DynamicMeanVFE.forward (detector3d/pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:38-76):
 *   floor((xyz-min)/voxel) -> in-range mask -> key=b*XYZ+x*YZ+y*Z+z -> torch.unique(sorted) -> scatter_mean.
 *   points: (P, point_stride) fp32 rows [batch_idx, x, y, z, extra...]; the first `num_features`
 *   columns after batch_idx are averaged.  Outputs are ordered by ascending key (same as torch.unique):
 *   voxel_coords (cap,4) int32 [b,z,y,x], voxel_features (cap,num_features), point_to_voxel (P) int32
 *   (row of the voxel each point fell in, -1 if out of range; may be NULL), *num_voxels (device int32).
 *   index_ws: sv_index_persistent_bytes(batch*X*Y*Z); scratch: sv_voxelize_dynamic_scratch_bytes().
 * ---------------------------------------------------------------------------------------------- */
size_t sv_voxelize_dynamic_scratch_bytes(int64_t num_points, int64_t ncells, int64_t capacity);
int sv_voxelize_dynamic(const float* points, int64_t num_points, int point_stride, int num_features,
                        const float* pc_range_host /*6*/, const float* voxel_size_host /*3*/,
                        const int32_t* grid_size_host /*3: X,Y,Z*/, int batch_size,
                        void* index_ws, void* scratch,
                        int32_t* voxel_coords, float* voxel_features, int32_t* point_to_voxel,
                        int64_t capacity, int32_t* num_voxels, void* stream);

/* MeanVFE.forward (backbones_3d/vfe/mean_vfe.py:14-31): sum over the point axis / clamp_min(count,1).
 * voxels (V, max_points, C) fp32, num_points (V) int32 -> out (V, C). */
int sv_mean_vfe(const float* voxels, const int32_t* num_points, int64_t num_voxels, int max_points,
                int num_features, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * VCN surface completion (see/surface_completion/models/vcn/models/VCN_VC.py:178-214, VCN_CN.py:142-156)
 *   Activations are channel-last (M = B*n points, C) fp32; weights keep PyTorch's (C_out, C_in) layout.
 * ---------------------------------------------------------------------------------------------- */
#define SV_ACT_NONE 0
#define SV_ACT_RELU 1
#define SV_ACT_LRELU 2

int sv_fill_f32(float* dst, int64_t n, float value, void* stream);
/* *out = mul * sum(x^2) and, y != NULL, y = scale * x in ONE pass over x (n % 4 == 0, 16-byte aligned): a mean-square loss over a dense tensor that
 * leaves its own gradient behind while it reads the tensor (mul = 1 / n, scale = 2 / n) -- bench.py's stand-in for the loss behind HeightCompression
 * (the reference's BEV backbone + head, base_bev_backbone.py / anchor_head_single.py, are not in the headline step).  Deterministic (fixed grid,
 * fixed-order sums); scratch: sv_mean_square_scratch_bytes().  sv_scale_by_device_scalar: x *= *g, skipped on the device when *g == 1. */
size_t sv_mean_square_scratch_bytes(void);
int sv_mean_square(const float* x, int64_t n, float mul, float scale, float* y, float* out, void* scratch, void* stream);
int sv_scale_by_device_scalar(float* x, int64_t n, const float* g, void* stream);

/* C[M,N] = act(A[M,K] @ W[N,K]^T + bias[N] + group_bias[row / rows_per_group][N])  on the fp32 MFMA
 * (v_mfma_f32_32x32x2_f32: exact fp32).  Replaces the Conv1d(k=1)(+BN folded)(+ReLU/LeakyReLU) layers of
 * pose_encoder / FeatureEncoder (VCN_VC.py:81-106,116-123) and the Linear layers (:124-131).
 * C may be NULL (no store); group_max (M/rows_per_group, N), pre-filled with -inf, receives the max over each
 * group of rows (torch.max(feature, dim=2), VCN_VC.py:100,104 and AdaptiveMaxPool1d :122).  K % 32 == 0. */
int sv_gemm_bias_act(const float* A, int lda, const float* W, int ldw, const float* bias,
                     const float* group_bias, int rows_per_group, float* C, int ldc, float* group_max,
                     int M, int N, int K, int act, float slope, void* stream);
/* The same product (uniform groups, C stored, no column max) with K split over blockIdx.z when the output has few 128 x 128 tiles and K is long --
 * the shared FC over the pooled RoI grid of PV-RCNN's head (pcdet/models/roi_heads/pvrcnn_head.py:44-61: 27 648 -> 256 on 512 RoIs: 8 tiles, 1.9 ms
 * in one pass).  The split sums are added in split order by a second launch (bitwise reproducible), which also applies bias / group bias /
 * activation.  sv_gemm_splitk_splits(M, N, K) = number of splits (1: the one-pass kernel runs); scratch: sv_gemm_splitk_scratch_bytes. */
int sv_gemm_splitk_splits(int M, int N, int K);
size_t sv_gemm_splitk_scratch_bytes(int M, int N, int K);
int sv_gemm_bias_act_splitk(const float* A, int lda, const float* W, int ldw, const float* bias, const float* group_bias, int rows_per_group,
                            float* C, int ldc, int M, int N, int K, int act, float slope, void* scratch, void* stream);

/* out[m][c] = act(weight[c][0..2] . xyz[m] + bias[c])   (the Conv1d(3, C, 1) first layers; C % 4 == 0) */
int sv_pointwise_conv3(const float* xyz, const float* weight, const float* bias, float* out, int64_t M,
                       int C, int act, float slope, void* stream);
/* The same layer on gathered rows: out[m] = f(xyz[sel[m]]) for m < *m_dev (<= M_cap), sel int64 row indices (sv_unique_rows_compact). */
int sv_pointwise_conv3_gather(const float* xyz, const int64_t* sel, int64_t M_cap, const int32_t* m_dev, const float* weight, const float* bias,
                              float* out, int C, int act, float slope, void* stream);

/* ---- training side of the dense per-point / per-RoI layer stacks (see-vcn_amd/csrc/dense_train.hip).  The reference runs these through cuDNN / cuBLAS
 * under autograd: VCN_VC.forward in training mode (see/surface_completion/models/vcn/models/VCN_VC.py:97-106,116-131,178-214) and PV-RCNN's
 * point head / feature fusion / RoI head FC stacks (dense_heads/point_head_simple.py, pfe/voxel_set_abstraction.py:168-172, roi_heads/pvrcnn_head.py:171-176).
 * sv_gemm_tn: C (N, K) = A^T B with A (M, N), B (M, K) row-major as they lie -- the weight gradient dW = dY^T X; fp32 MFMA, M split over workgroups,
 * partial tiles summed in a fixed order (bitwise reproducible).  scratch: sv_gemm_tn_scratch_bytes(M, N, K). */
size_t sv_gemm_tn_scratch_bytes(int64_t M, int N, int K);
int sv_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, void* scratch, void* stream);
/* C[i][j] = sum_c A[i * stride_a_row + c * stride_a_c] * B[c * stride_b_c + j * stride_b_col] for small / odd-shaped products (element strides);
 * contraction split over workgroups, fixed-order sum.  scratch: sv_gemm_strided_scratch_bytes(rows, cols, contraction). */
size_t sv_gemm_strided_scratch_bytes(int rows, int cols, int64_t contraction);
int sv_gemm_strided(const float* A, int64_t stride_a_row, int64_t stride_a_c, const float* B, int64_t stride_b_c, int64_t stride_b_col, float* C, int64_t ldc,
                    int rows, int cols, int64_t contraction, void* scratch, void* stream);
/* out[n] = sum_m A[m][n] (bias gradient), fixed order.  scratch: sv_column_sums_scratch_bytes(M, N). */
size_t sv_column_sums_scratch_bytes(int64_t M, int N);
int sv_column_sums(const float* A, int64_t lda, int64_t M, int N, float* out, void* scratch, void* stream);
/* Rows g * rows_per_group .. form group g (an object's points): column max with the first arg-max row (torch.max(dim) / AdaptiveMaxPool1d), its
 * backward (dx written everywhere: the gradient at the arg-max row, zero elsewhere) and the column sum of a group's rows (gradient of a feature
 * broadcast to the group's rows). */
int sv_segment_max(const float* x, int64_t ldx, int groups, int rows_per_group, int channels, float* out, int32_t* arg, void* stream);
int sv_segment_max_backward(const float* dout, const int32_t* arg, int groups, int rows_per_group, int channels, float* dx, int64_t ldx, void* stream);
int sv_segment_sum(const float* x, int64_t ldx, int groups, int rows_per_group, int channels, float* out, void* stream);
/* dz = dy * act'(y) for act = SV_ACT_RELU / SV_ACT_LRELU taken from the layer's output y (0: copy) */
int sv_act_backward(const float* dy, const float* y, int64_t n, int act, float slope, float* dz, void* stream);

/* VCN_VC.py:185-190: frustum angle, rotation to the frustum view, mean-centering.
 * input (B,n,3) -> fview (B,n,3), centred (B,n,3), state (B,32) [angle, mean xyz, centre xyz, rot 3x3]. */
int sv_vcn_vc_prep(const float* input, int batch, int n, float* fview, float* centred, float* state, void* stream);
/* VCN_VC.py:195-200: rel_pose (B,9) -> centre, rot (ortho6d, :36-49) into state; pc_cn = (fview-centre) @ rot^T */
int sv_vcn_vc_pose(const float* fview, int batch, int n, const float* rel_pose, float* state, float* pc_cn, void* stream);
/* VCN_VC.py:205-212: coarse_cn (B,nc,3) -> coarse (B,nc,3) in the sensor view, reg_rot (B,3,3), reg_centre (B,3) */
int sv_vcn_vc_finish(const float* coarse_cn, int batch, int num_coarse, const float* state, float* coarse,
                     float* reg_rot, float* reg_centre, void* stream);
/* VCN_CN.py:146-154 with utils/transform.py:91-160: inverse=0: vc_to_cn + normalize_scale; 1: restore_scale + cn_to_vc */
int sv_vcn_cn_transform(const float* in, int batch, int n, const float* gt_boxes, int inverse, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sparse 3-D convolution (replaces the un-vendored third-party `spconv`; call sites
 * detector3d/pcdet/models/backbones_3d/spconv_backbone.py:8-27,77-117,141-157, height_compression.py:21).
 * Coordinates are (N,4) int32 [b,z,y,x]; shapes/kernel/stride/padding/dilation are host int32[3] in z,y,x order.
 * Kernel offset index k = (kz*Ky + ky)*Kx + kx.  Rulebooks are output-major tables nbr[k][row] (or -1).
 * ---------------------------------------------------------------------------------------------- */
/* out_shape = floor((in + 2*pad - dil*(k-1) - 1)/stride) + 1 */
int sv_conv_out_shape(const int32_t* in_shape_host, const int32_t* ksize_host, const int32_t* stride_host,
                      const int32_t* padding_host, const int32_t* dilation_host, int32_t* out_shape_host);
/* scratch for either rulebook entry; `ncells` = batch * prod(shape of the level being indexed) */
size_t sv_rulebook_scratch_bytes(int64_t n_in, int64_t ncells);
/* SubMConv3d rulebook: nbr (K, n): row j with coord[j] = coord[i] + (k - K/2)*dilation (output set = input set,
 * caller's row order).  index_ws: sv_index_persistent_bytes(batch*Z*Y*X). dilation_host may be NULL (=1). */
int sv_rulebook_subm(const int32_t* coords, int64_t n, int batch, const int32_t* shape_host,
                     const int32_t* ksize_host, const int32_t* dilation_host, void* index_ws, void* scratch,
                     int32_t* nbr, void* stream);
/* Same table through a dense cell -> row map (int32 per cell in 4x4x8-cell tiles, sv_cellmap_persistent_bytes(batch, shape) bytes,
 * ALL ZERO on entry and on return): 3 launches instead of 8, no atomics.  For grids whose map fits comfortably in the 288 GB of HBM (16 KITTI scenes at
 * 5 cm: 5.9 GB); callers fall back to sv_rulebook_subm for larger grids. */
size_t sv_cellmap_persistent_bytes(int batch, const int32_t* spatial_shape);
/* table_rows (n, 32) int32 and masks (n) int32, both optional (K <= 27): the same table ROW-MAJOR ([0..26] source rows, rest -1: one 128-byte
 * line per row) and every row's neighbour mask (bit k = has neighbour k) -- what the convolution plan consumes (sv_conv_plan_build). */
int sv_rulebook_subm_cellmap(const int32_t* coords, int64_t n, int batch, const int32_t* spatial_shape, const int32_t* ksize,
                             const int32_t* dilation, void* cellmap, int32_t* nbr, int32_t* table_rows, int32_t* masks, void* stream);
/* SparseConv3d rulebook, phase 1: output coordinates in canonical (ascending ((b*Z+z)*Y+y)*X+x) order and the
 * input-major table nbr_in (K, n_in) = output row fed by (k, input) or -1; *num_out on the device.
 * in_block (optional, K <= 27): (32 + 1) * n_in int32 = [table_rows_in (n_in, 32) | masks_in (n_in)], the same table row-major plus every
 * input row's mask (bit k = feeds an output through offset k) -- what the data-gradient plan consumes.
 * index_ws: sv_index_persistent_bytes(batch * prod(out_shape)). */
int sv_rulebook_sparse(const int32_t* coords, int64_t n_in, int batch, const int32_t* in_shape_host,
                       const int32_t* ksize_host, const int32_t* stride_host, const int32_t* padding_host,
                       const int32_t* dilation_host, void* index_ws, void* scratch, int32_t* out_coords,
                       int32_t* nbr_in, int32_t* in_block, int64_t capacity, int32_t* num_out, void* stream);
/* A CHAIN of strided layers (spconv_backbone.py:141-157: conv2, conv3, conv4, conv_out; spconv reads every level's indice-pair count back
 * to the host) counted end to end on the device -- ONE device -> host read for all output-site counts, or none when the caller defers it:
 * level l marks its output cells from level l-1's site list (one lane per site; level 0 from `coords0`, whose length may live on the device:
 * n0_dev non-NULL overrides n0, which then only bounds the grid), counts its occupancy bitmap and writes its sites in canonical (ascending
 * key) order into sites[l] (capacity caps[l] rows of int4 [b,z,y,x]) and their number into num_out[l] (device).  Kernels up to 3 per axis.
 * geoms_host: n_levels x 15 int32 = {in_shape[3], ksize[3], stride[3], padding[3], dilation[3]}; index_ws[l]: sv_index_persistent_bytes(batch *
 * prod(out_shape_l)), all zero on entry and left MARKED (the SET jobs of sv_rulebook_batch return the touched words to zero);
 * scratch: sv_rulebook_chain_scratch_bytes(largest batch * prod(out_shape)). */
size_t sv_rulebook_chain_scratch_bytes(int64_t max_ncells);
int sv_rulebook_chain_count(const int32_t* coords0, int64_t n0, const int32_t* n0_dev, int batch, int n_levels, const int32_t* geoms_host,
                            void* const* index_ws, int32_t* const* sites, const int64_t* caps, int32_t* num_out, void* scratch, void* stream);
/* Every rulebook table of a network in three launches, through dense cell -> row maps (sv_cellmap_persistent_bytes per level, all zero between
 * calls).  jobs_host: n_jobs rows of 32 int64, row[0] = kind:
 *   1 SET    [1] sites (n, 4) of a level, [2] exact-size copy of them or 0, [3] the level's cell map or 0, [4] index workspace the sites were
 *            marked in (sv_rulebook_chain_count) to return to zero, or 0, [5] n, [6..8] level shape (Z, Y, X), [9] batch
 *   2 QUERY  one table: row r (coordinate c) and offset k = (kz, ky, kx) address cell (c * mul + add + k * step) / div of the TARGET level (exact
 *            division, inside its shape), whose map gives the source row.  [1] row sites, [2] target map, [3] nbr (K, n) k-major table, [4] (n, 32)
 *            row-major twin, [5] masks (n), [6] n, [7..9] row-level shape, [10..12] target shape, [13..15] ksize, [16..18] mul, [19..21] add,
 *            [22..24] step, [25..27] div, [28] batch.  Submanifold: mul 1, add -(ks/2)*dil, step dil, div 1; strided output-major: mul stride,
 *            add -pad, step dil, div 1 (target = input level); strided input-major: mul 1, add pad, step -dil, div stride (target = output level).
 * Runs every SET, then every QUERY, then zeroes the maps of the SET jobs.  Same tables as sv_rulebook_subm_cellmap / sv_rulebook_sparse +
 * sv_rulebook_invert_rows, bit for bit. */
int sv_rulebook_batch(const int64_t* jobs_host, int n_jobs, void* stream);
/* phase 2 (after the caller knows n_out): nbr_out (K, n_out) output-major table from nbr_in */
int sv_rulebook_invert(const int32_t* nbr_in, int64_t n_in, int K, int32_t* nbr_out, int64_t n_out, void* stream);
/* The same inversion (K <= 27) from the row-major input table (sv_rulebook_sparse's in_block) that also writes the row-major twin and the
 * neighbour masks of the output side:
 *   out_block ((32 + K + 1) * n_out int32, filled here) = [table_rows_out (n_out, 32) | nbr_out (K, n_out) | masks_out (n_out)] */
int sv_rulebook_invert_rows(const int32_t* table_rows_in, int64_t n_in, int K, int32_t* out_block, int64_t n_out, void* stream);
/* counts[k] = number of (in,out) pairs of offset k (spconv's indice_pair_num) */
int sv_rulebook_pair_counts(const int32_t* nbr, int64_t n_out, int K, int32_t* counts, void* stream);

/* Y[o][n] = epi( sum_k sum_c X[nbr[k][o]][c] * Wt[k][n][c] ),  epi: +bias, *scale+shift (folded BN), +residual, relu.
 *   forward:       X = features (N_in,C_in),  nbr = output-major table, Wt = weight as (K, C_out, C_in)
 *   backward-data: X = grad_out (N_out,C_out), nbr = input-major table, Wt = weight as (K, C_in, C_out)
 * X has n_src rows, nbr/Y have n_rows rows.  Plain entry: packed (K, Nc, Kd) weights and the k-major table; fp32 MFMA
 * (v_mfma_f32_16x16x4_f32, accumulators in registers) when Kd and Nc are multiples of 16, VALU otherwise (the 3-channel input layer).
 * This replaces spconv's per-offset gather / GEMM / scatter-add launches (spconv is un-vendored: spconv_backbone.py:8-27 are the call sites). */
int sv_sparse_conv_gather_gemm(const float* X, int64_t n_src, const int32_t* nbr, const float* Wt, float* Y,
                               int64_t n_rows, int K, int Kd, int Nc, const float* bias, const float* scale, const float* shift,
                               const float* residual, int relu, void* stream);

/* ---- the MFMA kernel on a PLAN of the table (what SubMConv3d / SparseConv3d of seevcn_amd.spconv run; csrc/sparse_conv.hip) ----
 * Inputs of a plan: the table row-major (n_rows, 32) and the rows' neighbour masks (n_rows) -- written by the rulebook builders
 * (sv_rulebook_subm_cellmap, sv_rulebook_invert_rows) or, for any k-major table with K <= 27, by sv_conv_table_rows.
 * sv_conv_plan_build, once per table: rows are split into 8 contiguous regions (one per XCD: workgroup b of the conv launch works on
 * region b % 8, so a scene's rows are gathered through one L2) and regrouped inside a region into 16-row tiles of equal neighbour-mask
 * class (counting sort): perm (16*ceil(n_rows/16)) int32 = row at each position (-1 = padding of the last tile), masks_p = its mask.
 * persistent: sv_conv_plan_persistent_bytes() bytes, all zero before the first call and left zero-consistent by every call.
 * sv_conv_plan_tiles, once per (table, tiles_per_wave): tile_of (sv_conv_plan_tiles_bytes(n_rows, tiles_per_wave)) maps [region][wave][slot] to a
 * tile of the region or -1.  A launch is one resident round of 4 waves per SIMD (8 regions x 128 workgroups); a region's tiles are
 * counting-sorted by their number of active offsets and dealt, round after round, to the 128 SIMDs of the region's XCD in ascending order of
 * their load so far (equal work per SIMD); a wave works through its slots tiles_per_wave tiles at a time.
 * tiles_per_wave = sv_conv_tiles_per_wave(n_rows, Kd, Nc) of the conv that will use it (2 or 4).
 * Results are bit-identical to sv_sparse_conv_gather_gemm (same summation order per output element). */
int sv_conv_table_rows(const int32_t* nbr, int64_t n_rows, int K, int32_t* table_rows, int32_t* masks, void* stream);
size_t sv_conv_plan_persistent_bytes(void);
size_t sv_conv_plan_perm_bytes(int64_t n_rows);
int sv_conv_plan_build(const int32_t* masks, int64_t n_rows, void* persistent, int32_t* perm, int32_t* masks_p, void* stream);
int sv_conv_tiles_per_wave(int64_t n_rows, int Kd, int Nc);
size_t sv_conv_plan_tiles_bytes(int64_t n_rows, int tiles_per_wave);
int sv_conv_plan_tiles(const int32_t* masks_p, int64_t n_rows, int tiles_per_wave, int32_t* tile_of, void* stream);
/* Both of the above for one tiles_per_wave value in ONE launch (one workgroup per region, all passes in LDS): perm, masks_p and tile_of as
 * sv_conv_plan_build + sv_conv_plan_tiles leave them, up to the order of the rows inside a class (results of the convolution do not depend
 * on it). */
int sv_conv_plan_build_dealt(const int32_t* masks, int64_t n_rows, int tiles_per_wave, int32_t* perm, int32_t* masks_p, int32_t* tile_of,
                             void* stream);
/* sv_conv_plan_build_dealt for several tables in ONE launch (one workgroup per region and table); jobs_host: n_jobs rows of 8 int64 =
 * {masks, n_rows, tiles_per_wave, perm, masks_p, tile_of, 0, 0} (device addresses). */
int sv_conv_plan_build_dealt_batch(const int64_t* jobs_host, int n_jobs, void* stream);
/* 1 iff the plan kernel is built for this layer shape (K <= 27 offsets, C_in in {16,32,64,128}, C_out in {16,32} or a multiple of 64 up to
 * 512) and X (n_src rows) is addressable through a 32-bit buffer descriptor; other shapes take sv_sparse_conv_gather_gemm. */
int sv_conv_mfma_kernel_applies(int K, int Kd, int Nc, int64_t n_src);
/* Weights of one layer in MFMA fragment order for both directions, from ANY (K, C_in, C_out) view given by its element strides (the
 * parameter of spconv-2.x layout (C_out, kz, ky, kx, C_in) is such a view: no transposing copy).  frag_fwd serves the forward
 * (Kd = C_in, Nc = C_out), frag_bwd the data gradient (Kd = C_out, Nc = C_in); K*C_in*C_out floats each, caller-owned (cache them per
 * weight version; either may be null). */
int sv_conv_weight_fragments(const float* W, int64_t stride_k, int64_t stride_cin, int64_t stride_cout, int K, int Cin, int Cout, float* frag_fwd,
                             float* frag_bwd, void* stream);
/* The same for n_layers weights in one launch.  descs_device: (n_layers, 10) int64 ON THE DEVICE, per layer {W pointer, stride_k, stride_cin,
 * stride_cout, K, C_in, C_out, frag_fwd pointer, frag_bwd pointer, first unit}; a layer has 2 * K * C_in * C_out / 4 units (float4, forward
 * then backward view), "first unit" is the running sum of the layers before it, total_units the sum over all layers. */
int sv_conv_weight_fragments_batch(const void* descs_device, int n_layers, int64_t total_units, void* stream);
/* Input transform of the NEXT sv_sparse_conv_gather_gemm_planned / sv_sparse_conv_wgrad* call of this host thread (consumed by it, whatever it
 * returns): that call reads X through y = [relu](x * scale[c] + shift[c]), coef (2, C_in) = scale | shift (16-byte aligned) -- X is then the RAW
 * output of the convolution below and coef the coefficients of its BatchNorm1d (sv_batchnorm_finalize_forward), i.e. the `norm_fn -> ReLU` tail of
 * post_act_block (spconv_backbone.py:9-27) is applied as the rows are gathered and the normalised tensor is never written.  Same expression as the
 * separate pass (fused multiply-add, then max): consumers see bit-identical values; absent neighbours contribute 0.  coef = NULL clears it.
 * Entry points that cannot apply it (the plain k-major conv, non-MFMA weight-gradient shapes) fail with SV_ERR_ARG. */
int sv_conv_next_input_norm(const float* coef, int relu);
/* table_k_reversed: offset k reads table entry K-1-k (a submanifold table serving its own data gradient, no flipped copy). */
int sv_sparse_conv_gather_gemm_planned(const float* X, int64_t n_src, const int32_t* table_rows, const int32_t* perm, const int32_t* masks_p,
                                       const int32_t* tile_of, int tiles_per_wave,
                                       const float* wfrag, float* Y, int64_t n_rows, int K, int Kd, int Nc, const float* bias,
                                       const float* scale, const float* shift, const float* residual, int relu, int table_k_reversed,
                                       float* bn_partial, void* stream);
/* Data gradient on a plan whose output is the gradient of a BatchNorm1d(+ReLU) OUTPUT (the layer below in VoxelBackBone8x, spconv_backbone.py:8-27):
 * the same kernel, and its epilogue also leaves the two per-channel sums of that BatchNorm's backward (sum of dy on the forward's ReLU branch,
 * and of dy * xhat) as sv_conv_planned_partials() per-workgroup partials in bn_partial -- sv_batchnorm_relu_backward_partial starts at the
 * combine.  bn_x = the BatchNorm's input (n_rows, Nc), bn_mean / bn_invstd its saved batch statistics, bn_gamma / bn_beta may be NULL. */
int sv_sparse_conv_dgrad_planned_bn(const float* dZ, int64_t n_src, const int32_t* table_rows, const int32_t* perm, const int32_t* masks_p,
                                    const int32_t* tile_of, int tiles_per_wave, const float* wfrag, float* dY, int64_t n_rows, int K, int Kd, int Nc,
                                    int table_k_reversed, const float* bn_x, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                    const float* bn_beta, int bn_relu, float* bn_partial, void* stream);
/* bn_partial (null or sv_conv_planned_partials() x 2 x Nc floats; plain epilogue only: no bias / scale / residual / relu): per-workgroup column
 * sums and sums of squares of Y, the first pass of the training-mode BatchNorm behind the convolution (post_act_block, spconv_backbone.py:9-27)
 * made in the epilogue that holds the values in registers anyway; consumed by sv_batchnorm_relu_forward_partial. */
int sv_conv_planned_partials(void);
/* Measurement aid (tools/conv_trace.py): while buf is non-null every wave of sv_sparse_conv_gather_gemm_planned writes 8 uint64 to it
 * (s_memtime at start / after the prologue / after the main loop / at the end, HW_ID, XCC_ID, tile-offset steps, block << 8 | wave);
 * buf holds grid.x * grid.y * 4 slots of 64 bytes (size it as 8 * (n_tiles + 64) * columns / 64 slots).  Not for production use. */
int sv_debug_conv_trace(void* buf);
/* The same for the MFMA weight gradient (tools/wgrad_trace.py; the 64 -> 64 channel instance only): per wave s_memtime at start, ticks spent in the
 * pass prologues (table read + compaction), ticks in the MFMA loops, s_memtime at the end, XCC_ID << 32 | HW_ID, s_memtime after the last pass,
 * passes << 32 | pairs, offset << 32 | chunk; buf holds 4 x workgroups slots of 64 bytes.  Not for production use. */
int sv_debug_wgrad_trace(void* buf);
/* dW (K, C_in, C_out) = sum_o X[nbr[k][o]]^T dY[o]; deterministic two-stage reduction.  n_src = rows of X (every table entry is < n_src): when
 * n_src * C_in * 4 < 2^32 the operand rows are addressed with 32-bit byte offsets from uniform bases; n_src <= 0 = unknown (64-bit addresses). */
size_t sv_sparse_conv_wgrad_scratch_bytes(int64_t n_rows, int K, int Cin, int Cout);
int sv_sparse_conv_wgrad(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K,
                         int Cin, int Cout, void* scratch, void* stream);
/* The same with element (k, c_in, c_out) written at dW[k * stride_k + c_in * stride_cin + c_out * stride_cout] -- the layout of the
 * caller's parameter (spconv's (C_out, kz, ky, kx, C_in)), so that no transposing copy of the gradient is needed.  The strides must
 * address a permutation of the K * C_in * C_out slab. */
int sv_sparse_conv_wgrad_strided(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K, int Cin,
                                 int Cout, int64_t stride_k, int64_t stride_cin, int64_t stride_cout, void* scratch, void* stream);
/* The two stages apart, for a backward pass that runs many layers: stage 1 (partial slabs into `partial`, sv_sparse_conv_wgrad_partial_bytes; *job = 10 int64
 * on the host describing the pending sum) per layer, then sv_sparse_conv_wgrad_reduce_batch(jobs, n_jobs) sums the slabs of all layers in one launch -- bitwise the
 * values sv_sparse_conv_wgrad_strided writes. */
size_t sv_sparse_conv_wgrad_partial_bytes(int64_t n_rows, int K, int Cin, int Cout);
int sv_sparse_conv_wgrad_stage1(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K, int Cin, int Cout,
                                int64_t stride_k, int64_t stride_cin, int64_t stride_cout, void* partial, int64_t* job, void* stream);
int sv_sparse_conv_wgrad_reduce_batch(const int64_t* jobs_host, int n_jobs, void* stream);
/* Weight gradient on EQUAL PIECES (the default of the trained path; replaces the role of spconv's indice_conv_backward filter gradient as the entries above do).
 * The chunked stage 1 above gives every (row chunk, offset) a workgroup: an offset's work follows its density, and the launch ends when the unluckiest CU
 * does (pairs per SIMD max / mean 1.4-1.6 on a LiDAR rulebook, tools/wgrad_trace.py).  Here a per-TABLE plan cuts the table's pairs, in (row eighth, offset, row) order, into
 * sv_wgrad_plan_pieces(Cin, Cout) pieces of equal pair count (up to one 64-row unit) -- as many as workgroups are resident -- and stage 1 runs one
 * workgroup per piece (one (Cin, Cout) slab per (row eighth, offset) a piece touches; the order of the cut is row eighth, offset, row, so that an XCD's
 * workgroups stay inside one eighth of the rows).  A plan is a function of (table, pieces) only: build it once per rulebook
 * table, use it for every layer on that table whose sv_wgrad_plan_pieces agrees.  Results are bitwise reproducible; they differ from the chunked
 * form's in the last bits (other partial sums).
 *   sv_wgrad_plan_bytes          bytes of a plan (device memory, caller-owned)
 *   sv_wgrad_plan_build          nbr (K, n_rows) -> plan; 2 launches
 *   sv_wgrad_planned_applies     1 iff the layer runs on this path (MFMA tile shape, 32-bit addressable operands); otherwise use sv_sparse_conv_wgrad*
 *   sv_sparse_conv_wgrad_planned_bytes   bytes of `partial`
 *   sv_sparse_conv_wgrad_planned         both stages; strides all 0 = contiguous (K, Cin, Cout) dW, else as sv_sparse_conv_wgrad_strided
 *   sv_sparse_conv_wgrad_planned_stage1  stage 1 only; *job for sv_sparse_conv_wgrad_reduce_batch */
size_t sv_wgrad_plan_bytes(int64_t n_rows, int K, int pieces);
int sv_wgrad_plan_pieces(int Cin, int Cout);
int sv_wgrad_plan_build(const int32_t* nbr, int64_t n_rows, int K, int pieces, void* plan, void* stream);
/* the plans of several tables in two launches; jobs_host: n_jobs rows of 8 int64 = {nbr, n_rows, K, pieces, plan, 0, 0, 0} (device addresses) */
int sv_wgrad_plan_build_batch(const int64_t* jobs_host, int n_jobs, void* stream);
int sv_wgrad_planned_applies(int64_t n_src, int64_t n_rows, int K, int Cin, int Cout);
size_t sv_sparse_conv_wgrad_planned_bytes(int K, int Cin, int Cout);
int sv_sparse_conv_wgrad_planned(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K, int Cin, int Cout,
                                 int64_t stride_k, int64_t stride_cin, int64_t stride_cout, const void* plan, void* partial, void* stream);
int sv_sparse_conv_wgrad_planned_stage1(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K, int Cin,
                                        int Cout, int64_t stride_k, int64_t stride_cin, int64_t stride_cout, const void* plan, void* partial, int64_t* job,
                                        void* stream);

/* SparseConvTensor.dense(): (N,C) + coords -> (B, C, D, H, W), every element written once */
size_t sv_sparse_to_dense_scratch_bytes(int batch, int D, int H, int W);
int sv_sparse_to_dense(const float* features, const int32_t* coords, int64_t n, int batch, int C, int D, int H, int W,
                       void* scratch, float* out, void* stream);
/* its backward: gather (B,C,D,H,W) at coords -> (N,C) */
int sv_dense_to_sparse(const float* dense, const int32_t* coords, int64_t n, int batch, int C, int D, int H, int W,
                       float* out, void* stream);
/* The same volume in channels-last memory for HeightCompression (height_compression.py:21-23: dense().view(N, C * D, H, W)) in front of a channels_last
 * 2-D backbone: out[b][y][x][c * D + d] -- on the torch side a (B, C D, H, W) tensor with channels_last strides -- and its backward from a gradient
 * in that order.  Scratch as sv_sparse_to_dense.  sv_sparse_to_dense_nhwc_applies: C % 4 == 0, D <= 8, H W % 16 == 0 (else use the pair above). */
int sv_sparse_to_dense_nhwc_applies(int C, int D, int H, int W);
int sv_sparse_to_dense_nhwc(const float* features, const int32_t* coords, int64_t n, int batch, int C, int D, int H, int W, void* scratch,
                            float* out, void* stream);
int sv_dense_to_sparse_nhwc(const float* dense, const int32_t* coords, int64_t n, int batch, int C, int D, int H, int W, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PointNet++ stacked-batch primitives (detector3d/pcdet/ops/pointnet2/pointnet2_stack/src/ *.cu; pybind names in
 * src/pointnet2_api.cpp:12-31).  Scenes are described by device int32 arrays of first row and row count.
 * ---------------------------------------------------------------------------------------------- */
/* farthest_point_sampling_wrapper(b, n, m, points, temp, idx) (src/sampling.cpp:24-36; kernel sampling_gpu.cu:24-140):
 * xyz (b,n,3) -> idx (b,m) int32, first index 0, index-exact including the reference's tie rule.
 * temp (b*n floats) is only used when n > 24576 (larger scenes stream from HBM); may be NULL otherwise. */
int sv_farthest_point_sampling(const float* xyz, int b, int n, int m, float* temp, int32_t* idx, void* stream);
/* the same over ragged scenes in one launch (replaces the per-scene Python loop of
 * VoxelSetAbstraction.get_sampled_points, backbones_3d/pfe/voxel_set_abstraction.py:250-256): idx (batch,m) holds
 * GLOBAL row indices (scene start + local index). max_n = largest scene. */
int sv_stack_farthest_point_sampling(const float* xyz, const int32_t* xyz_batch_start, const int32_t* xyz_batch_cnt, int batch,
                                     int max_n, int m, float* temp, int32_t* idx, void* stream);
/* The same with SEVERAL workgroups per scene (16; batch * 16 <= 256 workgroups, 4096 <= max_n <= 65536, m < 65536 -- otherwise it falls back
 * to one workgroup per scene): every round the workgroups exchange their best candidate through self-tagged 16-byte granules in
 * multi_scratch (sv_fps_multi_scratch_bytes(batch) bytes, any content).  Index-exact with the single-workgroup kernel.  This path reads an
 * error word back and therefore returns with the stream synchronised; if a partner workgroup never arrives (bounded poll) it retries with
 * write-through granules and then fails with SV_ERR_HIP.  SEEVCN_FPS_MULTI=0: always one workgroup per scene; =1: write-through only. */
size_t sv_fps_multi_scratch_bytes(int batch);
int sv_stack_farthest_point_sampling_multi(const float* xyz, const int32_t* xyz_batch_start, const int32_t* xyz_batch_cnt, int batch,
                                           int max_n, int m, float* temp, void* multi_scratch, int32_t* idx, void* stream);
/* The same without the read-back (seevcn extension, for a sampling that runs on a side stream beside the backbone): ONE attempt -- write_through 0:
 * records kept in one XCD's L2 (fast, relies on the observed dispatch order), 1: write-through records (any placement) -- and nothing is
 * synchronised.  The int32 error word at multi_scratch + sv_fps_multi_error_offset(batch) is 0 when every partner workgroup arrived; the
 * caller reads it when it next synchronises with `stream` and, if non-zero, samples again with write_through = 1. */
size_t sv_fps_multi_error_offset(int batch);
int sv_stack_farthest_point_sampling_multi_async(const float* xyz, const int32_t* xyz_batch_start, const int32_t* xyz_batch_cnt, int batch, int max_n,
                                                 int m, float* temp, void* multi_scratch, int32_t* idx, int write_through, void* stream);
/* The same samples from ONE workgroup per scene that skips what a pick cannot change (seevcn extension; csrc/fps_bucket.hip): the scene's points are
 * put in Morton-cell order (64 consecutive points = one bucket with a bounding box and its largest running distance), a round re-computes only the
 * buckets whose box is nearer to the new pick than that maximum -- an exact test, IEEE rounding being monotone -- and reduces the arg-max over the
 * per-bucket maxima.  Index-exact with sv_farthest_point_sampling / sv_stack_farthest_point_sampling including the tie rule; nothing to poll, nothing
 * to read back.  Layouts: starts / counts given: stacked scenes, idx (batch, m) GLOBAL rows; both NULL: `batch` scenes of fixed_n (= max_n) points, idx
 * scene-local.  sv_fps_bucket_applies: 2048 <= max_n <= 24576 and m > 1 (SEEVCN_FPS_BUCKET=0: never; SEEVCN_FPS_BUCKET_MIN overrides the lower bound);
 * scratch: sv_fps_bucket_scratch_bytes(batch, max_n) bytes, any content. */
int sv_fps_bucket_applies(int batch, int max_n, int m);
size_t sv_fps_bucket_scratch_bytes(int batch, int max_n);
int sv_farthest_point_sampling_bucketed(const float* xyz, const int32_t* xyz_batch_start, const int32_t* xyz_batch_cnt, int batch, int fixed_n, int max_n,
                                        int m, void* scratch, int32_t* idx, void* stream);
/* ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx) (src/ball_query.cpp:31-47,
 * kernel ball_query_gpu.cu:16-66): idx (M,nsample) scene-local indices of the first nsample points with d^2 < r^2 in index
 * order, padded with the first hit; idx[m][0] = -1 for an empty ball. */
int sv_ball_query_stack(int batch, int M, int max_queries_per_scene, float radius, int nsample, const float* new_xyz,
                        const int32_t* new_xyz_batch_start, const int32_t* new_xyz_batch_cnt, const float* xyz,
                        const int32_t* xyz_batch_start, const int32_t* xyz_batch_cnt, int32_t* idx, void* stream);
/* The same result, element for element, over a cell hash of the support points instead of the reference's scan of the whole scene per query
 * (seevcn extension): cells of edge radius * 1.0001 counting-sorted into hash buckets, a wave per query tests the candidates of the 27 cells
 * around it with the reference's own distance expression and writes the nsample smallest indices in ascending order.  n_points = rows of xyz;
 * scratch: sv_ball_query_hash_scratch_bytes(n_points) bytes (any content); nsample <= 64. */
size_t sv_ball_query_hash_scratch_bytes(int64_t n_points);
int sv_ball_query_stack_hashed(int batch, int M, int64_t n_points, float radius, int nsample, const float* new_xyz, const int32_t* new_xyz_batch_start,
                               const int32_t* new_xyz_batch_cnt, const float* xyz, const int32_t* xyz_batch_start, const int32_t* xyz_batch_cnt,
                               void* scratch, int32_t* idx, void* stream);
/* ---- fused set-abstraction reduction, eval mode (csrc/set_abstraction.hip): the tail of StackSAModuleMSG.forward
 * (ops/pointnet2/pointnet2_stack/pointnet2_modules.py:78-112) for one radius scale -- QueryAndGroup's gather (xyz relative to the query in
 * front of the features, an empty ball = zeros; pointnet2_utils.py:112-159) -> two Conv2d 1x1 + BatchNorm2d(eval) + ReLU -> max over nsample --
 * without the (M, C+3, nsample) tensor.  idx / row_start as for sv_group_points_stack (raw ball-query output).
 * sv_sa_prepare_weights folds an eval-mode BatchNorm into a conv weight (c_out, c_in): xyz_first = 1 for the first layer (c_in = 3 + C, C a
 * multiple of 16: output (c_out, 16*(C/16+1)) = [features | xyz | zeros]), 0 for the second (same shape); b_out (c_out).
 * sv_sa_mlp_max: C <= 128 features (multiple of 16, 0 = xyz only), nsample 16 or 32, C1 / C2 in {16, 32, 48, 64}; out (M, C2). */
int sv_sa_prepare_weights(const float* weight, const float* bn_weight, const float* bn_bias, const float* running_mean,
                          const float* running_var, float eps, int c_out, int c_in, int xyz_first, float* w_out, float* b_out, void* stream);
int sv_sa_mlp_max(const float* xyz, const float* features, const float* new_xyz, const int32_t* idx, const int32_t* row_start, int64_t M, int C,
                  int nsample, const float* w1, const float* b1, int C1, const float* w2, const float* b2, int C2, float* out, void* stream);
/* ---- set abstraction, TRAINING mode (csrc/set_abstraction_train.hip): one radius scale of StackSAModuleMSG.forward
 * (pointnet2_modules.py:78-112) with batch-statistics BatchNorm2d, and its backward, as chains of persistent MFMA launches over 16-row tiles;
 * no (M, C+3, nsample) tensor, no library GEMM.  Same operands as sv_sa_mlp_max, weights in the PARAMETER layouts: w1 (C1, 3 + C) with the xyz
 * columns first (QueryAndGroup's concatenation order, pointnet2_utils.py:147-152), w2 (C2, C1); gamma / beta / running_* / tracked = the two
 * nn.BatchNorm2d's weight, bias, running_mean, running_var, num_batches_tracked (updated like torch: biased variance to normalise, unbiased
 * into running_var).  R = M * nsample rows.
 * forward:  N = rows of xyz / features; proj (N, C1) is a work buffer (null when C == 0: the feature part of layer 1, made once per support
 *           point).  Writes z1 (R, C1), z2 (R, C2) [pre-BatchNorm activations, kept for the backward], save_mean* / save_invstd*, sel (M, C2) = the z2
 *           that makes each output, arg (M, C2) = its slot, out (M, C2); aux (M, C2) floats and aux_arg (M, C2) bytes are work buffers.
 * backward: grad_out (M, C2) -> grad_w1 (C1, 3 + C), grad_w2 (C2, C1), dgamma* / dbeta*, scatter (N, C1) = per support point the sum of dz1 over
 *           its (query, slot) pairs (zero-filled here; fp32 atomics like group_points_grad_kernel_stack, group_points_gpu.cu:38-41).  The
 *           feature gradient is linear in the gathered row: grad_features (N, C) = scatter . w1[:, 3:] (null when not wanted or C == 0).
 *           dy1 (R, C1) and aux (M, C2) are work buffers.
 * scratch: sv_sa_train_scratch_bytes(C, C1, C2) bytes, not shared between a forward and a backward in flight on different streams. */
size_t sv_sa_train_scratch_bytes(int C, int C1, int C2);
int sv_sa_train_forward(const float* xyz, const float* features, const float* new_xyz, const int32_t* idx, const int32_t* row_start, int64_t M,
                        int64_t N, int C, int nsample, const float* w1, const float* gamma1, const float* beta1, float* running_mean1, float* running_var1,
                        int64_t* tracked1, int C1, const float* w2, const float* gamma2, const float* beta2, float* running_mean2,
                        float* running_var2, int64_t* tracked2, int C2, float momentum, float eps, void* scratch, float* proj, float* z1,
                        float* z2,
                        float* save_mean1, float* save_invstd1, float* save_mean2, float* save_invstd2, float* sel, float* aux, uint8_t* arg,
                        uint8_t* aux_arg, float* out, void* stream);
int sv_sa_train_backward(const float* xyz, const float* features, const float* new_xyz, const int32_t* idx, const int32_t* row_start, int64_t M,
                         int64_t N, int C, int nsample, const float* w1, const float* gamma1, const float* beta1, int C1, const float* w2,
                         const float* gamma2, const float* beta2, int C2, const float* z1, const float* z2, const float* save_mean1,
                         const float* save_invstd1, const float* save_mean2, const float* save_invstd2, const float* sel, const uint8_t* arg,
                         const float* out, const float* grad_out, void* scratch, float* dy1, float* aux, float* scatter, float* grad_features,
                         float* grad_w1, float* grad_w2, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, void* stream);
/* group_points_wrapper / group_points_grad_wrapper (src/group_points.cpp:31-69, kernels group_points_gpu.cu:15-102):
 * out (M,C,nsample)[m][c][s] = features[row_start[m] + idx[m][s]][c]; row_start[m] = first feature row of query m's scene.
 * The gradient zero-fills grad_features (N,C) and scatter-adds with fp32 atomics like the reference. */
int sv_group_points_stack(int M, int C, int nsample, const float* features, const int32_t* idx, const int32_t* row_start,
                          float* out, void* stream);
int sv_group_points_grad_stack(int M, int C, int N, int nsample, const float* grad_out, const int32_t* idx,
                               const int32_t* row_start, float* grad_features, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rotated-box geometry (detector3d/pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:12-17,
 * detector3d/pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:172-177). Boxes are (N,7) fp32 [x,y,z,dx,dy,dz,heading].
 * ---------------------------------------------------------------------------------------------- */
/* boxes_overlap_bev_gpu (iou=0) / boxes_iou_bev_gpu (iou=1): out (num_a, num_b) */
int sv_boxes_overlap_bev(const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* out, int iou, void* stream);
/* boxes_iou3d_gpu (detector3d/pcdet/ops/iou3d_nms/iou3d_nms_utils.py:48-81: BEV overlap x height overlap / union volume, the elementwise chain
 * around boxes_overlap_bev_gpu) in ONE launch, for `batch` independent box sets at once: boxes_a (batch, num_a, stride_a), boxes_b (batch, num_b,
 * stride_b) rows whose first 7 floats are the box, out (batch, num_a, num_b). */
int sv_boxes_iou3d_batch(const float* boxes_a, int num_a, int stride_a, const float* boxes_b, int num_b, int stride_b, int batch, float* out,
                         void* stream);
/* nms_gpu (normal=0, rotated BEV IoU) / nms_normal_gpu (normal=1, axis-aligned) over boxes ALREADY sorted by score:
 * keep (n) int64 device indices of the kept boxes in order, *num_out device int32.  Unlike the reference
 * (iou3d_nms.cpp:111-131: D2H copy of the mask + host sweep) the greedy sweep runs on the device. n <= 65536. */
size_t sv_nms_scratch_bytes(int n);
int sv_nms(const float* boxes, int n, float thresh, int normal, void* scratch, int64_t* keep, int32_t* num_out, void* stream);
/* The same, stopping once max_keep boxes are kept: keep[0..*num_out) is the prefix sv_nms would return, *num_out = min(kept, max_keep)
 * (seevcn extension for callers that truncate to NMS_POST_MAXSIZE, model_nms_utils.py:21). */
int sv_nms_prefix(const float* boxes, int n, float thresh, int normal, int max_keep, void* scratch, int64_t* keep, int32_t* num_out, void* stream);
/* points_in_boxes_gpu: boxes (B,T,7), pts (B,M,3) -> out (B,M) int32 index of the first box containing the point or -1 */
int sv_points_in_boxes(const float* boxes, const float* pts, int batch, int num_boxes, int num_points, int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Anchor head (detector3d/pcdet/models/dense_heads/anchor_head_template.py, target_assigner/axis_aligned_target_assigner.py)
 * ---------------------------------------------------------------------------------------------- */
/* generate_predicted_boxes (anchor_head_template.py:225-272) with ResidualCoder.decode_torch (box_coder_utils.py:48-77):
 * anchors (A,7) shared by the batch, box_encodings (B,A,7), dir_cls_preds (B,A,num_dir_bins) or NULL -> out (B,A,7). */
int sv_anchor_decode(const float* anchors, int64_t num_anchors, const float* box_encodings, const float* dir_cls_preds,
                     int batch, int num_dir_bins, float dir_offset, float dir_limit_offset, float* out, void* stream);
/* AxisAlignedTargetAssigner.assign_targets (axis_aligned_target_assigner.py:36-210), deterministic branch (POS_FRACTION < 0,
 * MATCH_HEIGHT False, NORM_BY_NUM_EXAMPLES False): anchors (A,7) in head order [(z,y,x), set, size, rot]; set s owns the
 * per-location slots [set_offset[s], set_offset[s+1]) and matches ground truth whose class id (column 7, 1-based) is
 * set_class[s]; gt_boxes (B,G,8) zero-padded.  Outputs labels (B,A) int32 {-1 ignore, 0 background, class id},
 * reg_targets (B,A,7) (ResidualCoder.encode_torch), reg_weights (B,A).  gt_max_scratch: B*G floats.  G <= 128.
 * set_offset/set_class/thresholds are DEVICE arrays. */
int sv_assign_targets_axis_aligned(const float* anchors, int64_t num_anchors, int anchors_per_location, int num_sets,
                                   const int32_t* set_offset, const int32_t* set_class, const float* matched_thr,
                                   const float* unmatched_thr, const float* gt_boxes, int batch, int max_gt, float* gt_max_scratch,
                                   int32_t* labels, float* reg_targets, float* reg_weights, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Hard voxelisation + pillar features (PointPillars path)
 * ---------------------------------------------------------------------------------------------- */
/* spconv VoxelGenerator(V2) / Point2VoxelCPU3d semantics as called by DataProcessor.transform_points_to_voxels
 * (detector3d/pcdet/datasets/processor/data_processor.py:15-60,115-143): first-come points per voxel (<= max_points),
 * voxels numbered in order of first appearance (<= max_voxels).  One workgroup per scene.
 * points rows of point_stride floats, x at column xyz_offset, num_features columns copied from there;
 * scenes given by device arrays scene_start/scene_cnt.  Outputs per scene: voxels (B,max_voxels,max_points,C) zero-padded,
 * coords (B,max_voxels,3) [z,y,x], num_points_per_voxel (B,max_voxels), num_voxels (B). */
size_t sv_voxelize_hard_scratch_bytes(int batch, int64_t total_points, int max_scene_points);
int sv_voxelize_hard(const float* points, int point_stride, int xyz_offset, int num_features, const int32_t* scene_start,
                     const int32_t* scene_cnt, int batch, int64_t total_points, int max_scene_points,
                     const float* pc_range_host, const float* voxel_size_host, const int32_t* grid_size_host, int max_points,
                     int max_voxels, void* scratch, float* voxels, int32_t* coords, int32_t* num_points_per_voxel,
                     int32_t* num_voxels, void* stream);
/* PillarVFE.forward feature decoration (backbones_3d/vfe/pillar_vfe.py:94-118): (V,mp,C) -> (V,mp,C+6[+1]) */
int sv_pillar_decorate(const float* voxels, const int32_t* num_points, const int32_t* coords, int64_t num_voxels, int max_points,
                       int num_features, const float* voxel_size_host, const float* pc_range_host, int use_absolute_xyz,
                       int with_distance, float* out, void* stream);

/* interpolate_from_bev_features (detector3d/pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py:11-42,176-204):
 * keypoints (M,4) [b,x,y,z], bev (B,C,H,W) -> out (M,C); the gradient scatter-adds the taps into a channel-last staging map (scratch:
 * sv_bev_interpolate_grad_scratch_bytes bytes; a tap is then one contiguous run of C floats for the float atomics) and transposes it into
 * grad_bev (B,C,H,W), every element written. */
int sv_bev_interpolate(const float* keypoints, int64_t num_keypoints, const float* bev, int batch, int C, int H, int W, float x_min,
                       float y_min, float voxel_x, float voxel_y, float bev_stride, float* out, void* stream);
size_t sv_bev_interpolate_grad_scratch_bytes(int batch, int C, int H, int W);
int sv_bev_interpolate_grad(const float* keypoints, int64_t num_keypoints, const float* grad_out, int batch, int C, int H, int W,
                            float x_min, float y_min, float voxel_x, float voxel_y, float bev_stride, void* scratch, float* grad_bev, void* stream);

/* SigmoidFocalClassificationLoss.forward (detector3d/pcdet/utils/loss_utils.py:9-72: alpha-balanced sigmoid focal loss, un-reduced) and its
 * derivative w.r.t. the logits, one launch each instead of ~20 / ~30 elementwise ones.  input, target (n_rows, num_class), weights (n_rows) or
 * NULL.  grad_out NULL: out = loss (n_rows, num_class); grad_out given: out = grad_out * d loss / d input. */
int sv_sigmoid_focal_loss(const float* input, const float* target, const float* weights, int64_t n_rows, int num_class, float alpha, float gamma,
                          const float* grad_out, float* out, void* stream);
/* WeightedSmoothL1Loss.forward (loss_utils.py:75-136: code-wise weighted smooth-L1, nan targets ignored, un-reduced) and its derivative w.r.t.
 * input.  input, target (n_rows, num_codes), code_weights (num_codes) or NULL, weights (n_rows) or NULL; grad_out as above. */
int sv_weighted_smooth_l1_loss(const float* input, const float* target, const float* code_weights, const float* weights, int64_t n_rows,
                               int num_codes, float beta, const float* grad_out, float* out, void* stream);

/* CenterHead.assign_targets (detector3d/pcdet/models/dense_heads/center_head.py:103-213; gaussian_radius /
 * draw_gaussian_to_heatmap, models/model_utils/centernet_utils.py:9-69): all heads and scenes in one launch.
 * gt_boxes (B,G,box_dim) with the global 1-based class id in the last column (0 = padding); cls_to_local
 * (num_heads, num_class+1) device table = 1-based class index inside the head or 0.  Outputs: heatmaps (B,total_cls,H,W)
 * (head h owns channels [head_cls_offset[h], +head_num_class[h])), target_boxes (num_heads,B,num_max_objs,box_dim)
 * [dx, dy, z, log dims, cos, sin, extras], inds / masks (num_heads,B,num_max_objs) int64. */
int sv_center_assign_targets(const float* gt_boxes, int batch, int max_gt, int box_dim, int num_heads, int num_class,
                             const int32_t* cls_to_local, const int32_t* head_num_class, const int32_t* head_cls_offset, int total_cls,
                             int fm_w, int fm_h, float x_min, float y_min, float voxel_x, float voxel_y, float fm_stride,
                             int num_max_objs, float gaussian_overlap, int min_radius, float* heatmaps, float* target_boxes,
                             int64_t* inds, int64_t* masks, void* stream);

/* ---- pointnet2_batch_cuda wrappers (detector3d/pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:12-27; kernels in
 * ball_query_gpu.cu, group_points_gpu.cu, sampling_gpu.cu, interpolate_gpu.cu) and the stacked 3-NN interpolation
 * (pointnet2_stack/src/interpolate_gpu.cu:16-195).  Same argument order as the pybind wrappers, raw device pointers.
 * ball query: idx (B,m,nsample) must be zero-filled by the caller like the reference's Python does (queries without a hit
 * keep it).  Gradient entries zero-fill their output.  Batch farthest point sampling = sv_farthest_point_sampling. */
int sv_ball_query_batch(int batch, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int32_t* idx,
                        void* stream);
int sv_group_points_batch(int batch, int c, int n, int npoints, int nsample, const float* points, const int32_t* idx, float* out,
                          void* stream);
int sv_group_points_grad_batch(int batch, int c, int n, int npoints, int nsample, const float* grad_out, const int32_t* idx,
                               float* grad_points, void* stream);
int sv_gather_points_batch(int batch, int c, int n, int npoints, const float* points, const int32_t* idx, float* out, void* stream);
int sv_gather_points_grad_batch(int batch, int c, int n, int npoints, const float* grad_out, const int32_t* idx, float* grad_points,
                                void* stream);
int sv_three_nn_batch(int batch, int n, int m, const float* unknown, const float* known, float* dist2, int32_t* idx, void* stream);
int sv_three_interpolate_batch(int batch, int c, int m, int n, const float* points, const int32_t* idx, const float* weight, float* out,
                               void* stream);
int sv_three_interpolate_grad_batch(int batch, int c, int n, int m, const float* grad_out, const int32_t* idx, const float* weight,
                                    float* grad_points, void* stream);

/* ---- launch-list executor (csrc/sequencer.hip): enqueue a list of this library's operations with ONE call.  An operation is SV_OP_WORDS int64:
 *   [0] code   [1..8] eight small integers i0..i7   [9..12] four sizes / strides n0..n3   [13..16] four doubles (bit patterns) f0..f3
 *   [17..31] fifteen pointers p0..p14 (device addresses; 0 = null)
 * executed in order on `stream`; the first failing operation's code is returned (sv_last_error names it).  Codes and their fields, in the
 * argument order of the entry point each one calls:
 *   SV_OP_CONV_PLANNED  sv_sparse_conv_gather_gemm_planned: p = X, table_rows, perm, masks_p, tile_of, wfrag, Y, bias, scale, shift, residual,
 *                       bn_partial; n = n_src, n_rows; i = tiles_per_wave, K, Kd, Nc, relu, table_k_reversed
 *   SV_OP_CONV_PLAIN    sv_sparse_conv_gather_gemm: p = X, nbr, Wt, Y, bias, scale, shift, residual; n = n_src, n_rows; i = K, Kd, Nc, relu
 *   SV_OP_BN_FWD        sv_batchnorm_relu_forward (i3 = 0) / _forward_partial (i3 = number of partials): p = x, gamma, beta, running_mean,
 *                       running_var, scratch, y, save_mean, save_invstd, num_batches_tracked; n = rows; i = channels, training, relu, n_partials;
 *                       f = momentum, eps
 *   SV_OP_BN_BWD        sv_batchnorm_relu_backward (i2 = 0) / _backward_partial (i2 = number of partials): p = x, dy, gamma, beta, save_mean,
 *                       save_invstd, scratch, dx, dgamma, dbeta; n = rows; i = channels, relu, n_partials
 *   SV_OP_WGRAD         sv_sparse_conv_wgrad_strided: p = X, nbr, dY, dW, scratch; n = n_rows, stride_k, stride_cin, stride_cout; i = K, Cin, Cout, n_src;
 *                       p5 = equal-pieces plan of the table (then sv_sparse_conv_wgrad_planned, p4 = sv_sparse_conv_wgrad_planned_bytes) or 0
 *   SV_OP_WGRAD_DEFERRED  the same fields as SV_OP_WGRAD, but only stage 1 runs in place (sv_sparse_conv_wgrad_stage1; p4 = this layer's OWN partial region of
 *                       sv_sparse_conv_wgrad_partial_bytes) and the slabs of every deferred layer are summed by ONE launch at the end of the list
 *                       (sv_sparse_conv_wgrad_reduce_batch): the gradients are complete when sv_run_ops returns, bitwise the values of SV_OP_WGRAD
 *   SV_OP_DGRAD_PLANNED_BN  sv_sparse_conv_dgrad_planned_bn: p = dZ, table_rows, perm, masks_p, tile_of, wfrag, dY, bn_x, bn_mean, bn_invstd, bn_gamma,
 *                       bn_beta, bn_partial; n = n_src, n_rows; i = tiles_per_wave, K, Kd, Nc, table_k_reversed, bn_relu
 *   SV_OP_BN_FINALIZE   sv_batchnorm_finalize_forward: p = gamma, beta, running_mean, running_var, scratch, coef, save_mean, save_invstd,
 *                       num_batches_tracked, x (0: partials in scratch); n = rows; i = channels, n_partials; f = momentum, eps
 *   SV_OP_BN_APPLY      sv_batchnorm_apply: p = x, coef, y; n = rows; i = channels, relu
 * Input transform: SV_OP_CONV_PLANNED with p12 = coef (2, Kd) and i6 = relu, SV_OP_WGRAD[_DEFERRED] with p6 = coef (2, Cin) and i4 = relu read their
 * X through sv_conv_next_input_norm(coef, relu): X is then the RAW output of the convolution below, its BatchNorm (+ReLU) is applied as the rows are
 * gathered, and the normalised activation tensor is never written.
 * Used by seevcn_amd/spconv/chain.py: the forward and the backward of a conv -> BatchNorm -> ReLU chain (VoxelBackBone8x, spconv_backbone.py:128-180)
 * as two calls inside one autograd node. */
#define SV_OP_WORDS 32
#define SV_OP_CONV_PLANNED 1
#define SV_OP_CONV_PLAIN 2
#define SV_OP_BN_FWD 3
#define SV_OP_BN_BWD 5
#define SV_OP_WGRAD 6
#define SV_OP_DGRAD_PLANNED_BN 7
#define SV_OP_WGRAD_DEFERRED 8
#define SV_OP_BN_FINALIZE 9
#define SV_OP_BN_APPLY 10
int sv_run_ops(const int64_t* ops, int n_ops, void* stream);
/* measurement form: events around every operation on `stream`, the call waits for the stream and writes each operation's elapsed milliseconds to
 * ms[0 .. n_ops) (bench.py's roofline block times the conv launches of the step's own launch lists with it) */
int sv_run_ops_timed(const int64_t* ops, int n_ops, void* stream, float* ms);
/* The same list with its weight gradients (SV_OP_WGRAD, SV_OP_WGRAD_DEFERRED, the deferred reduction) on `side_stream`: each goes behind an event recorded on
 * `stream` after the operations in front of it; the other operations do not wait for it; `stream` waits for `side_stream` once, at the end, so that when
 * the call returns everything is ordered on `stream` as after sv_run_ops.  A weight gradient only feeds the optimiser: the backward chain need not stop
 * for it, and its matrix-core work fills the bandwidth-bound BatchNorm launches and the tails of the data-gradient launches.  Same kernels, same
 * results. */
int sv_run_ops_two_streams(const int64_t* ops, int n_ops, void* stream, void* side_stream);

/* ---- BatchNorm1d (+ReLU) on (N,C) voxel features: the norm_fn -> ReLU tail of post_act_block
 * (detector3d/pcdet/models/backbones_3d/spconv_backbone.py:9-27,73; torch.nn.BatchNorm1d semantics: biased batch variance for
 * normalisation, unbiased for running_var, running = (1-momentum)*running + momentum*batch).  C multiple of 4, C/4 divides 256.
 * scratch: sv_batchnorm_scratch_bytes(C) bytes (uninitialised).  gamma/beta may be null (affine=False); running_* may be null when training.
 * Backward recomputes the ReLU mask from x, so only x, save_mean and save_invstd need to be kept.  num_batches_tracked (device
 * int64, may be null): nn.BatchNorm1d's counter, += 1 by the training forward. */
size_t sv_batchnorm_scratch_bytes(int channels);
int sv_batchnorm_relu_forward(const float* x, int64_t n, int channels, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float momentum, float eps, int training, int relu, void* scratch, float* y,
                              float* save_mean, float* save_invstd, int64_t* num_batches_tracked, void* stream);
/* the training forward with its statistics pass already done by the producer of x: scratch holds n_partials x 2 x channels partial sums behind its
 * 4 * channels coefficient floats (sv_sparse_conv_gather_gemm_planned's bn_partial = (float*)scratch + 4 * channels) */
int sv_batchnorm_relu_forward_partial(const float* x, int64_t n, int channels, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, float momentum, float eps, int relu, void* scratch, int n_partials, float* y,
                                      float* save_mean, float* save_invstd, int64_t* num_batches_tracked, void* stream);
/* the statistics of the training forward WITHOUT its elementwise pass: partials in scratch as above -> save_mean, save_invstd, running statistics and
 * coef (2, channels) = scale | shift of y = x * scale + shift in the caller's buffer (16-byte aligned); y itself is made by the consumers as they read
 * x (sv_conv_next_input_norm) or by sv_batchnorm_apply where a tensor is needed */
int sv_batchnorm_finalize_forward(const float* x /* NULL: partials in scratch; else the statistics pass over x runs first */, int64_t n, int channels, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  float momentum, float eps, void* scratch, int n_partials, float* coef, float* save_mean, float* save_invstd,
                                  int64_t* num_batches_tracked, void* stream);
/* y = [relu](x * scale + shift), coef (2, channels) = scale | shift: the elementwise pass on its own */
int sv_batchnorm_apply(const float* x, int64_t n, int channels, const float* coef, int relu, float* y, void* stream);
int sv_batchnorm_relu_backward(const float* x, const float* dy, int64_t n, int channels, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, int relu, void* scratch, float* dx, float* dgamma,
                               float* dbeta, void* stream);
/* sv_batchnorm_relu_backward with the first pass of its two sums already in `scratch` (n_partials workgroup partials behind the 4 * C coefficient
 * floats, written by sv_sparse_conv_dgrad_planned_bn's epilogue): combine + elementwise pass only. */
int sv_batchnorm_relu_backward_partial(const float* x, const float* dy, int64_t n, int channels, const float* gamma, const float* beta,
                                       const float* save_mean, const float* save_invstd, int relu, void* scratch, int n_partials, float* dx,
                                       float* dgamma, float* dbeta, void* stream);

/* Ragged-group variant of sv_gemm_bias_act: row_group[M] (non-decreasing int32) names the group of every row; used after
 * sv_unique_rows, when each object keeps only its distinct points (ResamplePoints, vcn/datasets/data_transforms.py:254-262,
 * tiles Ni points to 1024 copies: every Conv1d(k=1) row and the max-pools over them only depend on the distinct rows). */
int sv_gemm_bias_act_ragged(const float* A, int lda, const float* W, int ldw, const float* bias, const float* group_bias,
                            const int32_t* row_group, float* C, int ldc, float* group_max, int M, int N, int K, int act, float slope,
                            void* stream);
/* The ragged form with the number of rows ON THE DEVICE: M_cap rows of A / C / row_group are addressable and size the launch, the first *m_dev
 * (<= M_cap) are computed -- sv_unique_rows_compact's total feeds it directly, so that VCN's forward needs no device -> host read. */
int sv_gemm_bias_act_ragged_dev(const float* A, int lda, const float* W, int ldw, const float* bias, const float* group_bias,
                                const int32_t* row_group, float* C, int ldc, float* group_max, int M_cap, const int32_t* m_dev, int N, int K,
                                int act, float slope, void* stream);
/* x (B,n,3), n <= 1024 -> uniq_idx (B,n): the first counts[b] entries of row b index one copy of each distinct point of object b
 * (exact float equality, -0.0 == 0.0), lexicographic order. */
int sv_unique_rows(const float* x, int batch, int n, int32_t* uniq_idx, int32_t* counts, void* stream);
/* Packs sv_unique_rows' per-object lists: sel (>= sum(counts)) int64 flat row b*n + uniq_idx[b][r], row_group int32 = b, object
 * after object; *total (device int32) = sum(counts).  Replaces ~25 torch indexing kernels in the VCN eval forward. */
int sv_unique_rows_compact(const int32_t* uniq_idx, const int32_t* counts, int batch, int n, int64_t* sel, int32_t* row_group,
                           int32_t* total, void* stream);

/* ---- Chamfer distance of the VCN training loss: the reference's `chamfer` extension,
 * see/surface_completion/models/vcn/extensions/chamfer_dist/chamfer_cuda.cpp:36-39 (forward -> [dist1, dist2, idx1, idx2],
 * backward -> [grad_xyz1, grad_xyz2]) over chamfer.cu:15-201.  xyz1 (B,n,3), xyz2 (B,m,3); squared distances; idx = first
 * nearest neighbour; backward zero-fills its outputs. */
int sv_chamfer_forward(const float* xyz1, const float* xyz2, int batch, int n, int m, float* dist1, float* dist2, int32_t* idx1,
                       int32_t* idx2, void* stream);
int sv_chamfer_backward(const float* xyz1, const float* xyz2, const int32_t* idx1, const int32_t* idx2, const float* grad_dist1,
                        const float* grad_dist2, int batch, int n, int m, float* grad_xyz1, float* grad_xyz2, void* stream);

/* ---- VCN post-processing (CPU code in the reference) ------------------------------------------
 * partial_with_KDTree / get_partial_mesh_batch (see/surface_completion/models/vcn/utils/sampling.py:8-41,69-81): per object,
 * np.unique(partial) -> k nearest coarse points each (float64 distances) -> list(set(indices)) in CPython's set iteration
 * order -> np.tile(...)[:surface_pts].  partial (B,n,3), complete (B,m,3), n,m <= 1024 -> surface (B,surface_pts,3),
 * n_selected (B) = size of the index set. */
size_t sv_vcn_surface_select_scratch_bytes(int batch);
int sv_vcn_surface_select(const float* partial, const float* complete, int batch, int n_partial, int n_complete, int k,
                          int surface_pts, void* scratch, float* surface, int32_t* n_selected, void* stream);
/* get_largest_cluster(_batch) (sampling.py:83-110): open3d cluster_dbscan(eps, min_points) + np.argmax(np.bincount(labels>=0))
 * + tile to total_pts.  min_points <= 2 (VCN.inference passes 2, models/VCN.py:90-93).  n_cluster[b] = 0 when every point is
 * noise (the reference raises there); out rows of that object are left untouched. */
int sv_vcn_largest_cluster(const float* points, int batch, int n, double eps, int min_points, int total_pts, float* out,
                           int32_t* n_cluster, void* stream);
/* the same with a hint (device int32 per object): the cloud of object b repeats its first period[b] rows cyclically -- what sv_vcn_surface_select returns
 * (np.tile(selected)[:surface_pts], sampling.py:37-39, n_selected as the period).  The pair tests then run over the first period[b] rows only (copies hang
 * under their originals: distance 0); the hint is verified on the device, a cloud that does not repeat takes the all-pairs path.  Same output. */
int sv_vcn_largest_cluster_periodic(const float* points, int batch, int n, const int32_t* period, double eps, int min_points, int total_pts, float* out,
                                    int32_t* n_cluster, void* stream);
/* Exact duplicates among n rows [b,x,y,z] (float32, 16-byte aligned): every copy of a row but the first gets b = -1 in place.  The unsorted form of np.unique(np.vstack(instances), axis=0) (SEE_VCN.py:115) for consumers that need the set only; no host
 * sync.  scratch: sv_dedup_rows_scratch_bytes(n) bytes, any content. */
size_t sv_dedup_rows_scratch_bytes(int64_t n);
int sv_dedup_rows(float* rows, int64_t n, void* scratch, void* stream);

/* replace_with_completed_pts (see/surface_completion/SEE_VCN.py:247-265): near[i] = 1 iff some ref point lies closer than
 * thresh to query i (float64 distance, strict <).  ref is expected row-sorted (np.unique) for the tile culling to pay.
 * row_dim 3: rows [x,y,z]; row_dim 4: rows [b,x,y,z] (a batch of scenes, only rows with equal b are compared). */
int sv_points_near_set(const float* query, int64_t n_query, const float* ref, int64_t n_ref, int row_dim, double thresh,
                       uint8_t* near, void* stream);
/* The same result through per-tile boxes made once (scratch: sv_points_near_set_scratch_bytes(n_ref) bytes): waves open a reference tile
 * only if one of their queries is within thresh of its box; reference rows with scene id < 0 (row_dim 4, sv_dedup_rows) are ignored. */
size_t sv_points_near_set_scratch_bytes(int64_t n_ref);
int sv_points_near_set_boxed(const float* query, int64_t n_query, const float* ref, int64_t n_ref, int row_dim, double thresh, void* scratch,
                             uint8_t* near, void* stream);

/* ---- point isolation: the CPU step in front of VCN --------------------------------------------------
 * isolate_gt_pts (see/surface_completion/SEE_VCN.py:61-82): open3d `pcd.crop(OrientedBoundingBox)` for every box of a scene.
 * boxes (G,15) float64 host-or-device rows [centre(3), R row-major(9), extent(3)] (device pointer); a point is inside iff
 * |(p - c) . R[:,a]| <= extent[a]/2 for a = 0,1,2 (float64, like open3d's GetPointIndicesWithinBoundingBox).
 * out_index (G,cap): ascending point indices of box g (first cap of them); out_count (G): their number (may exceed cap). */
int sv_crop_points_in_boxes(const float* points, int64_t n_points, int row_stride, const double* boxes, int n_boxes, int64_t cap,
                            int32_t* out_index, int32_t* out_count, void* stream);
/* KittiObjects.map_pointcloud_to_image (datasets/kitti/kitti_objects.py:153-176) with Calibration.project_velo_to_imageuv /
 * project_velo_to_rect (datasets/kitti/kitti_utils.py:69-114), float64.  v2c (3x4), r0 (3x3), p (3x4): HOST pointers, row-major.
 * fov[i] = 0 <= u < img_w and 0 <= v < img_h and x > min_dist; uv (n,2) = floor(u,v) (-1 outside the FOV); rect (n,3) optional. */
int sv_project_lidar_to_image_kitti(const float* points, int64_t n_points, int row_stride, const double* v2c, const double* r0,
                                    const double* p, int img_w, int img_h, double min_dist, int32_t* uv, uint8_t* fov, float* rect,
                                    void* stream);
/* CustomDatasetObjects.map_pointcloud_to_image (datasets/custom_dataset/custom_dataset_objects.py:141-192): extrinsic (3x4), intrinsic
 * (3x3), distcoeff (5) HOST pointers row-major; camera_model 0 = "pinhole" (k1,k2,p1,p2,k3), 1 = "equidistant" (4 coefficients).
 * uvd_int (n,3) = round-half-even [u, v, depth] (-1 outside), uvd (n,3) float64 optional, fov[i] = z_cam > 0 and |x/z| < atan(W/H) and
 * 0 < u < W-1 and 0 < v < H-1. */
int sv_project_lidar_to_image_camera(const float* points, int64_t n_points, int row_stride, const double* extrinsic, const double* intrinsic,
                                     const double* distcoeff, int camera_model, int img_w, int img_h, int32_t* uvd_int, double* uvd,
                                     uint8_t* fov, void* stream);
/* NuScenesObjects.map_pointcloud_to_image (datasets/nuscenes/nuscenes_objects.py:237-295; the nuscenes-devkit chain it restates, devkit not
 * vendored): lidar -> ego(sweep) -> global -> ego(image) -> camera with float32 storage after each of the 8 rotate / translate steps, then
 * view_points(K, normalize) in float64.  rotations (4,3,3) HOST row-major: [R_lidar2ego, R_ego2global, R_ego_cam2global^T, R_cam2ego^T];
 * translations (4,3) HOST: [t_lidar, t_ego, -t_ego_cam, -t_cam]; intrinsic (3,3) HOST.  fov[i] = depth > min_dist and 0 < u < W and 0 < v < H;
 * pts_img (n,2) = floor(u,v) (-1 outside the FOV); pc_cam (n,3) float32 camera-frame points. */
int sv_project_lidar_to_image_nuscenes(const float* points, int64_t n_points, int row_stride, const double* rotations, const double* translations,
                                       const double* intrinsic, int img_w, int img_h, double min_dist, float* pc_cam, int32_t* pts_img,
                                       uint8_t* fov, void* stream);
/* get_pts_in_mask (datasets/shared_utils.py:36-106): per instance the FOV points with mask[v,u] set.  Give either masks
 * (I,img_h,img_w) uint8 or rects (I,4) int32 [x0,y0,x1,y1] (use_bbox, :56-60).  Lists as in sv_crop_points_in_boxes. */
/* COCO polygons -> binary instance masks on the device: `dataset.annToMask(instance)` of get_pts_in_mask (shared_utils.py:66), i.e. pycocotools'
 * rleFrPoly + rleMerge(union) + rleDecode (cocoapi common/maskApi.c) with the same boundary arithmetic; the run-length code is replaced by a parity
 * scan (a pixel is inside iff an odd number of run boundaries lies at or before its column-major index).  xy: flat (x, y) doubles of all polygons;
 * poly_off (n_polygons + 1) first vertex of each; poly_inst (n_polygons) the instance each belongs to; masks (n_instances, img_h, img_w) uint8, written
 * whole; scratch: sv_polygon_masks_scratch_bytes.  max_vertices = the longest polygon (<= 4096); img_w <= 8192. */
size_t sv_polygon_masks_scratch_bytes(int n_polygons, int img_h, int img_w);
int sv_polygons_to_masks(const double* xy, const int32_t* poly_off, const int32_t* poly_inst, int n_polygons, int max_vertices, int n_instances, int img_h,
                         int img_w, void* scratch, uint8_t* masks, void* stream);
/* The same with every polygon's boundary moved inwards first (the reference's shrink_instance_masks, shared_utils.py:295-330: Polygon.buffer(-d) with
 * d = SHRINK_MASK_PERCENTAGE % of the half diagonal of the polygon's bounding box, then annToMask): a pixel of the polygon's mask is kept when its
 * centre lies at least shrink[p] from every edge of polygon p.  This is the region GEOS's negative buffer describes, sampled at the pixel centres --
 * NOT its vertex list (arcs cut into 16 chords per quadrant, vertices truncated to int, rasterised again): masks agree except in a band of about one
 * pixel along the shrunken boundary (shapely is not available here: unpinned).  shrink NULL = no shrinking; kept (n_polygons int32, nullable): pixels
 * each polygon wrote, 0 = its shrunken part is empty (the caller then falls back like shared_utils.py:325-326). */
int sv_polygons_to_masks_shrunk(const double* xy, const int32_t* poly_off, const int32_t* poly_inst, const double* shrink, int n_polygons, int max_vertices,
                                int n_instances, int img_h, int img_w, void* scratch, uint8_t* masks, int32_t* kept, void* stream);
int sv_points_in_masks(const int32_t* uv, const uint8_t* fov, int64_t n_points, const uint8_t* masks, const int32_t* rects,
                       int n_instances, int img_w, int img_h, int64_t cap, int32_t* out_index, int32_t* out_count, void* stream);
/* isolate_det_pts (SEE_VCN.py:144-181) and db_scan(..., return_largest_cluster=True) (shared_utils.py:395-409), one workgroup
 * per instance: eps = clip(eps_scaling * (|mean(xyz)| * tan_vres), min_eps, max_eps) unless fixed_eps >= 0; open3d
 * cluster_dbscan(eps, min_points) incl. border points; first largest cluster.  Instance g = counts[g] points: rows
 * point_index[starts[g]+r] of `points` (or rows starts[g]+r when point_index is null).  out_local[starts[g]+r] = position within
 * the instance of the r-th member (ascending), out_count[g] = members (0: instance has <= min_cluster points or is all noise;
 * -1: counts[g] > max_points), out_eps[g] = eps used.  Instances above 4096 points need
 * sv_isolate_cluster_scratch_bytes(n_instances, max_points) bytes of scratch. */
int64_t sv_isolate_cluster_scratch_bytes(int n_instances, int64_t max_points);
int sv_isolate_largest_cluster(const float* points, int row_stride, const int32_t* point_index, const int64_t* starts,
                               const int32_t* counts, int n_instances, int64_t max_points, double tan_vres, double eps_scaling,
                               double min_eps, double max_eps, double fixed_eps, int min_points, int min_cluster, void* scratch,
                               int32_t* out_local, int32_t* out_count, double* out_eps, void* stream);

/* ---- roiaware_pool3d_cuda.points_in_boxes_cpu (detector3d/pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-165; a host loop in the
 * reference): out (N,M) int32 0/1, box test with MARGIN 1e-2 (points_in_boxes_gpu uses 1e-5 and returns one box per point).
 * The RoI-aware pooling entries of that module (forward / backward: PartA2 only) and the PV-RCNN++ entries of pointnet2_stack_cuda
 * (voxel_query, vector_pool, local 3-NN) are not on the path of any BASELINE config and are not part of this library. */
int sv_points_in_boxes_matrix(const float* boxes, const float* pts, int num_boxes, int num_points, int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEEVCN_HIP_H */
