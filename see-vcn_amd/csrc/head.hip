// Anchor head: fused box decode (+ direction-bin fix-up) and fused axis-aligned target assignment.
//
// Reference (a chain of ~25 small elementwise torch kernels per call, an (A,G) IoU matrix materialised per class,
// python loops over batch x class):
//   AnchorHeadTemplate.generate_predicted_boxes   detector3d/pcdet/models/dense_heads/anchor_head_template.py:225-272
//   ResidualCoder.decode_torch / encode_torch     detector3d/pcdet/utils/box_coder_utils.py:13-77
//   limit_period                                  detector3d/pcdet/utils/common_utils.py:22-25
//   AxisAlignedTargetAssigner.assign_targets      dense_heads/target_assigner/axis_aligned_target_assigner.py:36-210
//   boxes3d_nearest_bev_iou / boxes_iou_normal    detector3d/pcdet/utils/box_utils.py:286-335
// HBM-bound: decode moves 4*(7 anchors + 7 deltas + NB dir + 7 out) bytes per anchor = 92*A for NB=2.
#include <math.h>

#include "common.h"

constexpr float PI_F = 3.14159265358979323846f;

// ------------------------------------------------------------------------------------------------ decode
__global__ __launch_bounds__(256) void k_anchor_decode(int64_t total, int64_t A, const float* __restrict__ anchors, const float* __restrict__ enc,
                                                       const float* __restrict__ dir_logits, int num_bins, float dir_offset,
                                                       float dir_limit_offset, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float* an = anchors + (i % A) * 7;
    const float* t = enc + i * 7;
    const float xa = an[0], ya = an[1], za = an[2], dxa = an[3], dya = an[4], dza = an[5], ra = an[6];
    const float diag = sqrtf(dxa * dxa + dya * dya);
    float* o = out + i * 7;
    o[0] = t[0] * diag + xa;
    o[1] = t[1] * diag + ya;
    o[2] = t[2] * dza + za;
    o[3] = expf(t[3]) * dxa;
    o[4] = expf(t[4]) * dya;
    o[5] = expf(t[5]) * dza;
    float rg = t[6] + ra;
    if (dir_logits) {
      const float* d = dir_logits + i * num_bins;
      int lab = 0;
      float best = d[0];
      for (int k = 1; k < num_bins; ++k)
        if (d[k] > best) { best = d[k]; lab = k; }               // first maximum, like torch.max(dim=-1)[1] on ties
      const float period = 2.0f * PI_F / (float)num_bins;
      const float val = rg - dir_offset;
      const float rot = val - floorf(val / period + dir_limit_offset) * period;   // limit_period
      rg = rot + dir_offset + period * (float)lab;
    }
    o[6] = rg;
  }
}

extern "C" int sv_anchor_decode(const float* anchors, int64_t num_anchors, const float* box_encodings, const float* dir_cls_preds,
                                int batch, int num_dir_bins, float dir_offset, float dir_limit_offset, float* out, void* stream) {
  SV_CHECK_ARG(num_anchors >= 0 && batch >= 0, "anchor_decode: bad arguments");
  const int64_t total = num_anchors * batch;
  if (total == 0) return SV_OK;
  SV_CHECK_ARG(anchors && box_encodings && out && (!dir_cls_preds || num_dir_bins >= 1), "anchor_decode: null pointer");
  hipLaunchKernelGGL(k_anchor_decode, dim3(sv_grid_1d(total, 256)), dim3(256), 0, sv_stream(stream), total, num_anchors, anchors, box_encodings,
                     dir_cls_preds, num_dir_bins, dir_offset, dir_limit_offset, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------ target assignment
struct AlignedBox { float x1, y1, x2, y2; };

__device__ __forceinline__ AlignedBox nearest_bev(const float* b) {          // box_utils.py:312-323
  const float r = b[6];
  const float rot = fabsf(r - floorf(r / PI_F + 0.5f) * PI_F);
  const bool keep = rot < PI_F / 4;
  const float d0 = keep ? b[3] : b[4], d1 = keep ? b[4] : b[3];
  return {b[0] - d0 / 2, b[1] - d1 / 2, b[0] + d0 / 2, b[1] + d1 / 2};
}

__device__ __forceinline__ float iou_aligned(const AlignedBox& a, const AlignedBox& b) {   // box_utils.py:286-309
  const float xl = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.f);
  const float yl = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.f);
  const float inter = xl * yl;
  const float area_a = (a.x2 - a.x1) * (a.y2 - a.y1), area_b = (b.x2 - b.x1) * (b.y2 - b.y1);
  return inter / fmaxf(area_a + area_b - inter, 1e-6f);
}

struct AssignArgs {
  const float* anchors;        // (A,7) in output order [(z,y,x), set, size, rot]
  const float* gt;             // (B,G,8) [box7, class]
  int64_t A;
  int B, G, per_loc, num_sets;
  const int* set_offset;       // (num_sets+1) prefix of anchors per location per set
  const int* set_class;        // (num_sets) 1-based class id matched by the set's anchors
  const float* matched_thr;    // (num_sets)
  const float* unmatched_thr;  // (num_sets)
  float* gt_max;               // (B,G) scratch
  int32_t* labels;             // (B,A)
  float* targets;              // (B,A,7)
  float* reg_weights;          // (B,A)
};

constexpr int AS_MAX_GT = 128;

__device__ __forceinline__ int anchor_set(const AssignArgs& a, int64_t ai) {
  const int o = (int)(ai % a.per_loc);
  int s = 0;
  while (s + 1 < a.num_sets && o >= a.set_offset[s + 1]) ++s;
  return s;
}

// grid.y = scene; GT boxes of the scene staged in LDS
template <int PASS>
__global__ __launch_bounds__(256) void k_assign(AssignArgs a) {
  __shared__ AlignedBox sg[AS_MAX_GT];
  __shared__ int scls[AS_MAX_GT];
  __shared__ float sgmax[AS_MAX_GT];
  __shared__ float sraw[AS_MAX_GT][7];
  const int b = blockIdx.y;
  for (int g = threadIdx.x; g < a.G; g += blockDim.x) {
    const float* p = a.gt + ((int64_t)b * a.G + g) * 8;
    sg[g] = nearest_bev(p);
    scls[g] = (int)p[7];
    for (int k = 0; k < 7; ++k) sraw[g][k] = p[k];
    if (PASS == 1) {
      const float m = a.gt_max[(int64_t)b * a.G + g];
      sgmax[g] = m == 0.f ? -1.f : m;                               // empty_gt_mask -> -1 (:152-153)
    } else {
      sgmax[g] = 0.f;
    }
  }
  __syncthreads();
  for (int64_t ai = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ai < a.A; ai += (int64_t)gridDim.x * blockDim.x) {
    const float* an = a.anchors + ai * 7;
    const int s = anchor_set(a, ai);
    const int cls = a.set_class[s];
    const AlignedBox ab = nearest_bev(an);
    float best = -1.f;
    int arg = -1;
    bool force = false;
    for (int g = 0; g < a.G; ++g) {
      if (scls[g] != cls) continue;
      const float v = iou_aligned(ab, sg[g]);
      if (PASS == 0) {
        if (v > 0.f) atomicMax(reinterpret_cast<int*>(&sgmax[g]), __float_as_int(v));   // IoU >= 0: int order = float order
      } else {
        if (v > best) { best = v; arg = g; }                         // first maximum (argmax on CPU)
        force = force || (v == sgmax[g]);
      }
    }
    if (PASS == 1) {
      const int64_t o = (int64_t)b * a.A + ai;
      int label;
      if (arg < 0) {
        label = 0;                                                   // no ground truth of this class: all background (:183-184)
      } else {
        const bool pos = best >= a.matched_thr[s];
        label = (force || pos) ? cls : (best < a.unmatched_thr[s] ? 0 : -1);
      }
      a.labels[o] = label;
      float* t = a.targets + o * 7;
      if (label > 0) {
        // ResidualCoder.encode_torch (box_coder_utils.py:13-46)
        const float* gtb = sraw[arg];
        const float dxa = fmaxf(an[3], 1e-5f), dya = fmaxf(an[4], 1e-5f), dza = fmaxf(an[5], 1e-5f);
        const float dxg = fmaxf(gtb[3], 1e-5f), dyg = fmaxf(gtb[4], 1e-5f), dzg = fmaxf(gtb[5], 1e-5f);
        const float diag = sqrtf(dxa * dxa + dya * dya);
        t[0] = (gtb[0] - an[0]) / diag;
        t[1] = (gtb[1] - an[1]) / diag;
        t[2] = (gtb[2] - an[2]) / dza;
        t[3] = logf(dxg / dxa);
        t[4] = logf(dyg / dya);
        t[5] = logf(dzg / dza);
        t[6] = gtb[6] - an[6];
        a.reg_weights[o] = 1.0f;
      } else {
        for (int k = 0; k < 7; ++k) t[k] = 0.f;
        a.reg_weights[o] = 0.f;
      }
    }
  }
  if (PASS == 0) {
    __syncthreads();
    for (int g = threadIdx.x; g < a.G; g += blockDim.x)
      if (sgmax[g] > 0.f) atomicMax(reinterpret_cast<int*>(&a.gt_max[(int64_t)b * a.G + g]), __float_as_int(sgmax[g]));
  }
}

extern "C" int sv_assign_targets_axis_aligned(const float* anchors, int64_t num_anchors, int anchors_per_location, int num_sets,
                                              const int32_t* set_offset, const int32_t* set_class, const float* matched_thr,
                                              const float* unmatched_thr, const float* gt_boxes, int batch, int max_gt, float* gt_max_scratch,
                                              int32_t* labels, float* reg_targets, float* reg_weights, void* stream) {
  SV_CHECK_ARG(num_anchors >= 0 && batch >= 0 && max_gt >= 0 && num_sets >= 1 && anchors_per_location >= 1, "assign_targets: bad arguments");
  SV_CHECK_ARG(max_gt <= AS_MAX_GT, "assign_targets: at most %d ground-truth boxes per scene", AS_MAX_GT);
  if (num_anchors == 0 || batch == 0) return SV_OK;
  SV_CHECK_ARG(anchors && set_offset && set_class && matched_thr && unmatched_thr && labels && reg_targets && reg_weights &&
                   (max_gt == 0 || (gt_boxes && gt_max_scratch)),
               "assign_targets: null pointer");
  hipStream_t st = sv_stream(stream);
  AssignArgs a{anchors, gt_boxes, num_anchors, batch, max_gt, anchors_per_location, num_sets, set_offset, set_class, matched_thr, unmatched_thr,
               gt_max_scratch, labels, reg_targets, reg_weights};
  dim3 grid(sv_grid_1d(num_anchors, 256, 512), batch);
  if (max_gt > 0) {
    SV_HIP(hipMemsetAsync(gt_max_scratch, 0, (size_t)batch * max_gt * 4, st));
    hipLaunchKernelGGL(k_assign<0>, grid, dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(k_assign<1>, grid, dim3(256), 0, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// Bilinear interpolation of BEV features at keypoints (interpolate_from_bev_features + bilinear_interpolate_torch,
// detector3d/pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py:11-42,176-204).  The reference permutes every scene's
// (C,H,W) map to (H,W,C) (a full copy) and gathers with advanced indexing in a python loop over the batch; here one launch
// reads the NCHW map in place.  keypoints (M,4) [b,x,y,z] -> out (M,C).
// ------------------------------------------------------------------------------------------------
struct BevGeom { float x0, y0, vx, vy, stride; int B, C, H, W; };

__device__ __forceinline__ void bev_taps(const float* kp, const BevGeom& g, int& b, int& x0, int& x1, int& y0, int& y1, float& wa, float& wb,
                                         float& wc, float& wd) {
  b = (int)kp[0];
  const float x = ((kp[1] - g.x0) / g.vx) / g.stride, y = ((kp[2] - g.y0) / g.vy) / g.stride;
  const int fx0 = (int)floorf(x), fy0 = (int)floorf(y);
  x0 = min(max(fx0, 0), g.W - 1); x1 = min(max(fx0 + 1, 0), g.W - 1);
  y0 = min(max(fy0, 0), g.H - 1); y1 = min(max(fy0 + 1, 0), g.H - 1);
  wa = ((float)x1 - x) * ((float)y1 - y);      // weights use the CLAMPED corners, like the reference (:36-39)
  wb = ((float)x1 - x) * (y - (float)y0);
  wc = (x - (float)x0) * ((float)y1 - y);
  wd = (x - (float)x0) * (y - (float)y0);
}

__global__ __launch_bounds__(256) void k_bev_interp(const float* __restrict__ kps, int64_t M, const float* __restrict__ bev, BevGeom g,
                                                    float* __restrict__ out) {
  const int64_t total = M * g.C;
  const int64_t hw = (int64_t)g.H * g.W;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = e / g.C;
    const int c = (int)(e - m * g.C);
    int b, x0, x1, y0, y1; float wa, wb, wc, wd;
    bev_taps(kps + m * 4, g, b, x0, x1, y0, y1, wa, wb, wc, wd);
    float v = 0.f;
    if (b >= 0 && b < g.B) {
      const float* p = bev + ((int64_t)b * g.C + c) * hw;
      v = p[(int64_t)y0 * g.W + x0] * wa + p[(int64_t)y1 * g.W + x0] * wb + p[(int64_t)y0 * g.W + x1] * wc + p[(int64_t)y1 * g.W + x1] * wd;
    }
    out[e] = v;
  }
}

// gradient of k_bev_interp.  The map is (B, C, H, W): the C values of one tap are H * W floats apart, and a float atomic whose 64 lanes hit 64
// different rows runs at 1/17 of the rate of one that adds 256 contiguous bytes (MI355X_MICROARCH.md, global float atomics) -- 0.93 ms for 16 384
// keypoints x 4 taps x 256 channels when done in place.  So the taps are added into a channel-last staging map (B, H, W, C), where a tap is
// one contiguous run of C floats, and a tiled transpose then writes the (B, C, H, W) gradient: every element exactly once, no memset of it.
__global__ __launch_bounds__(256) void k_bev_interp_grad_nhwc(const float* __restrict__ kps, int64_t M, const float* __restrict__ gout, BevGeom g,
                                                              float* __restrict__ stage) {
  const int64_t total = M * g.C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = e / g.C;
    const int c = (int)(e - m * g.C);
    int b, x0, x1, y0, y1; float wa, wb, wc, wd;
    bev_taps(kps + m * 4, g, b, x0, x1, y0, y1, wa, wb, wc, wd);
    if (b < 0 || b >= g.B) continue;
    float* p = stage + (int64_t)b * g.H * g.W * g.C + c;
    const float go = gout[e];
    atomicAdd(&p[((int64_t)y0 * g.W + x0) * g.C], go * wa);
    atomicAdd(&p[((int64_t)y1 * g.W + x0) * g.C], go * wb);
    atomicAdd(&p[((int64_t)y0 * g.W + x1) * g.C], go * wc);
    atomicAdd(&p[((int64_t)y1 * g.W + x1) * g.C], go * wd);
  }
}

// (B, HW, C) -> (B, C, HW), 64 x 64 tiles through LDS: reads run along C, writes along HW
__global__ __launch_bounds__(256) void k_nhwc_to_nchw(const float* __restrict__ in, int64_t HW, int C, float* __restrict__ out) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* src = in + (int64_t)b * HW * C;
  float* dst = out + (int64_t)b * C * HW;
  for (int r = ty; r < 64; r += 4) {
    const int64_t p = p0 + r;
    tile[r][tx] = (p < HW && c0 + tx < C) ? src[p * C + c0 + tx] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r;
    const int64_t p = p0 + tx;
    if (c < C && p < HW) dst[(int64_t)c * HW + p] = tile[tx][r];
  }
}

extern "C" int sv_bev_interpolate(const float* keypoints, int64_t num_keypoints, const float* bev, int batch, int C, int H, int W, float x_min,
                                  float y_min, float voxel_x, float voxel_y, float bev_stride, float* out, void* stream) {
  SV_CHECK_ARG(num_keypoints >= 0 && batch > 0 && C > 0 && H > 0 && W > 0, "bev_interpolate: bad arguments");
  if (num_keypoints == 0) return SV_OK;
  SV_CHECK_ARG(keypoints && bev && out, "bev_interpolate: null pointer");
  BevGeom g{x_min, y_min, voxel_x, voxel_y, bev_stride, batch, C, H, W};
  hipLaunchKernelGGL(k_bev_interp, dim3(sv_grid_1d(num_keypoints * C, 256, 256 * 16)), dim3(256), 0, sv_stream(stream), keypoints, num_keypoints,
                     bev, g, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" size_t sv_bev_interpolate_grad_scratch_bytes(int batch, int C, int H, int W) { return (size_t)batch * C * H * W * sizeof(float); }

// scratch: sv_bev_interpolate_grad_scratch_bytes(batch, C, H, W) bytes (the channel-last staging map)
extern "C" int sv_bev_interpolate_grad(const float* keypoints, int64_t num_keypoints, const float* grad_out, int batch, int C, int H, int W,
                                       float x_min, float y_min, float voxel_x, float voxel_y, float bev_stride, void* scratch, float* grad_bev,
                                       void* stream) {
  SV_CHECK_ARG(num_keypoints >= 0 && batch > 0 && C > 0 && H > 0 && W > 0 && grad_bev && scratch, "bev_interpolate_grad: bad arguments");
  hipStream_t st = sv_stream(stream);
  if (num_keypoints == 0) {
    SV_HIP(hipMemsetAsync(grad_bev, 0, (size_t)batch * C * H * W * 4, st));
    return SV_OK;
  }
  SV_CHECK_ARG(keypoints && grad_out, "bev_interpolate_grad: null pointer");
  float* stage = reinterpret_cast<float*>(scratch);
  SV_HIP(hipMemsetAsync(stage, 0, (size_t)batch * C * H * W * 4, st));
  BevGeom g{x_min, y_min, voxel_x, voxel_y, bev_stride, batch, C, H, W};
  hipLaunchKernelGGL(k_bev_interp_grad_nhwc, dim3(sv_grid_1d(num_keypoints * C, 256, 256 * 16)), dim3(256), 0, st, keypoints, num_keypoints, grad_out, g,
                     stage);
  const int64_t hw = (int64_t)H * W;
  hipLaunchKernelGGL(k_nhwc_to_nchw, dim3((unsigned)((hw + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)batch), dim3(256), 0, st, stage, hw, C, grad_bev);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// CenterHead target assignment (CenterHead.assign_targets / assign_target_of_single_head,
// detector3d/pcdet/models/dense_heads/center_head.py:103-213 with gaussian_radius / draw_gaussian_to_heatmap,
// models/model_utils/centernet_utils.py:9-69).  The reference loops over heads x scenes x boxes in python on CPU tensors
// (`.cpu()` at :204) and copies every map back; here one launch covers all heads and scenes: a workgroup per (head, scene)
// ranks the head's boxes in order (slot k = k-th box of the head), splats the Gaussians with an order-independent atomic max
// and writes inds / masks / regression targets.
// ------------------------------------------------------------------------------------------------
struct CenterArgs {
  const float* gt;            // (B, G, box_dim) last column = global class id (1-based), 0 = padding
  int B, G, box_dim, num_heads, num_class;
  const int32_t* cls_to_local;  // (num_heads, num_class + 1): local class id (1-based) of a global class in that head or 0
  const int32_t* head_ncls;     // (num_heads) classes per head
  const int32_t* head_cls_off;  // (num_heads) first heatmap channel of the head in the concatenated output
  int total_cls;
  int W, H;                   // feature map size (x, y)
  float x0, y0, vx, vy, stride;
  int num_max_objs, min_radius;
  float min_overlap;
  float* heatmap;             // (B, total_cls, H, W)  zero-filled by the launcher
  float* target_boxes;        // (num_heads, B, num_max_objs, box_dim)    [dx,dy offset, z, log dims(3), cos, sin, extras...]
  int64_t* inds;              // (num_heads, B, num_max_objs)
  int64_t* masks;             // (num_heads, B, num_max_objs)
};

__device__ __forceinline__ float gaussian_radius_f(float height, float width, float min_overlap) {   // centernet_utils.py:9-37
  const float b1 = height + width;
  const float c1 = width * height * (1 - min_overlap) / (1 + min_overlap);
  const float r1 = (b1 + sqrtf(b1 * b1 - 4 * 1 * c1)) / 2;
  const float b2 = 2 * (height + width);
  const float c2 = (1 - min_overlap) * width * height;
  const float r2 = (b2 + sqrtf(b2 * b2 - 4 * 4 * c2)) / 2;
  const float a3 = 4 * min_overlap, b3 = -2 * min_overlap * (height + width), c3 = (min_overlap - 1) * width * height;
  const float r3 = (b3 + sqrtf(b3 * b3 - 4 * a3 * c3)) / 2;
  return fminf(fminf(r1, r2), r3);
}

__global__ __launch_bounds__(256) void k_center_targets(CenterArgs a) {
  __shared__ int s_count;
  __shared__ int wcnt[4];
  const int head = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int tb_dim = a.box_dim;                       // box_dim - 1 (class) + 1 (heading split into cos, sin)
  float* tbox = a.target_boxes + ((int64_t)head * a.B + b) * a.num_max_objs * tb_dim;
  int64_t* inds = a.inds + ((int64_t)head * a.B + b) * a.num_max_objs;
  int64_t* masks = a.masks + ((int64_t)head * a.B + b) * a.num_max_objs;
  for (int e = tid; e < a.num_max_objs * tb_dim; e += 256) tbox[e] = 0.f;
  for (int e = tid; e < a.num_max_objs; e += 256) { inds[e] = 0; masks[e] = 0; }
  if (tid == 0) s_count = 0;
  __syncthreads();
  // rank the head's boxes in their original order: slot k = number of earlier boxes of this head (center_head.py:136)
  for (int g0 = 0; g0 < a.G; g0 += 256) {
    const int g = g0 + tid;
    int local = 0;
    if (g < a.G) {
      const int cls = (int)a.gt[((int64_t)b * a.G + g) * a.box_dim + a.box_dim - 1];
      if (cls >= 1 && cls <= a.num_class) local = a.cls_to_local[head * (a.num_class + 1) + cls];
    }
    const unsigned long long m = __ballot(local > 0);
    if (lane == 0) wcnt[wid] = __popcll(m);
    __syncthreads();
    int base = s_count;
    for (int w = 0; w < wid; ++w) base += wcnt[w];
    // process this chunk's boxes
    if (g < a.G && local > 0) {
      const int k = base + __popcll(m & ((1ull << lane) - 1ull));
      if (k < a.num_max_objs) {
        const float* bx = a.gt + ((int64_t)b * a.G + g) * a.box_dim;
        float cx = (bx[0] - a.x0) / a.vx / a.stride, cy = (bx[1] - a.y0) / a.vy / a.stride;
        cx = fminf(fmaxf(cx, 0.f), (float)a.W - 0.5f);
        cy = fminf(fmaxf(cy, 0.f), (float)a.H - 0.5f);
        const int ix = (int)cx, iy = (int)cy;
        const float dx = bx[3] / a.vx / a.stride, dy = bx[4] / a.vy / a.stride;
        if (dx > 0.f && dy > 0.f && ix >= 0 && ix <= a.W && iy >= 0 && iy <= a.H) {
          int radius = (int)gaussian_radius_f(dx, dy, a.min_overlap);
          radius = max(radius, a.min_radius);
          // draw_gaussian_to_heatmap (centernet_utils.py:48-69): sigma = diameter/6, window clipped to the map
          const double sigma = (double)(2 * radius + 1) / 6.0;
          const int left = min(ix, radius), right = min(a.W - ix, radius + 1), top = min(iy, radius), bottom = min(a.H - iy, radius + 1);
          float* hm = a.heatmap + (((int64_t)b * a.total_cls + a.head_cls_off[head] + (local - 1)) * a.H) * a.W;
          for (int yy = -top; yy < bottom; ++yy)
            for (int xx = -left; xx < right; ++xx) {
              const double h = exp(-(double)(xx * xx + yy * yy) / (2.0 * sigma * sigma));
              const float v = h < 2.220446049250313e-16 ? 0.f : (float)h;
              atomicMax(reinterpret_cast<int*>(&hm[(int64_t)(iy + yy) * a.W + ix + xx]), __float_as_int(v));
            }
          inds[k] = (int64_t)iy * a.W + ix;
          masks[k] = 1;
          float* t = tbox + (int64_t)k * tb_dim;
          t[0] = cx - (float)ix; t[1] = cy - (float)iy; t[2] = bx[2];
          t[3] = logf(bx[3]); t[4] = logf(bx[4]); t[5] = logf(bx[5]);
          t[6] = cosf(bx[6]); t[7] = sinf(bx[6]);
          for (int e = 7; e < a.box_dim - 1; ++e) t[e + 1] = bx[e];
        }
      }
    }
    __syncthreads();
    if (tid == 0) s_count += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
}

extern "C" int sv_center_assign_targets(const float* gt_boxes, int batch, int max_gt, int box_dim, int num_heads, int num_class,
                                        const int32_t* cls_to_local, const int32_t* head_num_class, const int32_t* head_cls_offset, int total_cls,
                                        int fm_w, int fm_h, float x_min, float y_min, float voxel_x, float voxel_y, float fm_stride,
                                        int num_max_objs, float gaussian_overlap, int min_radius, float* heatmaps, float* target_boxes,
                                        int64_t* inds, int64_t* masks, void* stream) {
  SV_CHECK_ARG(batch > 0 && max_gt >= 0 && box_dim >= 8 && num_heads > 0 && fm_w > 0 && fm_h > 0 && num_max_objs > 0, "center_assign_targets: bad arguments");
  SV_CHECK_ARG(cls_to_local && head_num_class && head_cls_offset && heatmaps && target_boxes && inds && masks && (max_gt == 0 || gt_boxes),
               "center_assign_targets: null pointer");
  hipStream_t st = sv_stream(stream);
  SV_HIP(hipMemsetAsync(heatmaps, 0, (size_t)batch * total_cls * fm_h * fm_w * 4, st));
  CenterArgs a{gt_boxes, batch, max_gt, box_dim, num_heads, num_class, cls_to_local, head_num_class, head_cls_offset, total_cls, fm_w, fm_h,
               x_min, y_min, voxel_x, voxel_y, fm_stride, num_max_objs, min_radius, gaussian_overlap, heatmaps, target_boxes, inds, masks};
  hipLaunchKernelGGL(k_center_targets, dim3(num_heads, batch), dim3(256), 0, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// Loss functions of the heads as one launch per direction.  The reference evaluates them as chains of elementwise torch ops
// (detector3d/pcdet/utils/loss_utils.py:9-136: 20 launches forward and ~30 in autograd's backward for the focal loss, 10 + 15 for the smooth-L1)
// on tensors of a few hundred thousand elements: launch-bound on any GPU.  Same arithmetic here, operation by operation in the forward; the
// backward is the analytic derivative w.r.t. the prediction (targets and weights are constants, as in the reference's use).
// ------------------------------------------------------------------------------------------------------------------------------------------
struct FocalTerm {
  float alpha_w, pt, p, bce;
};
__device__ __forceinline__ FocalTerm focal_terms(float x, float t, float alpha) {
  FocalTerm f;
  f.p = 1.f / (1.f + expf(-x));                                        // torch.sigmoid
  f.alpha_w = t * alpha + (1.f - t) * (1.f - alpha);
  f.pt = t * (1.f - f.p) + (1.f - t) * f.p;
  f.bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));             // max(x, 0) - x z + log(1 + exp(-|x|))
  return f;
}
template <bool GRAD>
__global__ __launch_bounds__(256) void k_sigmoid_focal(int64_t total, int C, const float* __restrict__ x, const float* __restrict__ t,
                                                        const float* __restrict__ w, float alpha, float gamma, const float* __restrict__ gout,
                                                        float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float xv = x[i], tv = t[i], wv = w ? w[i / C] : 1.f;
    const FocalTerm f = focal_terms(xv, tv, alpha);
    const float mod = gamma == 2.f ? f.pt * f.pt : powf(f.pt, gamma);  // torch.pow(pt, 2.0) is a square
    if (!GRAD) {
      out[i] = f.alpha_w * mod * f.bce * wv;
    } else {
      // d/dx [alpha_w pt^gamma bce] = alpha_w (gamma pt^(gamma-1) pt' bce + pt^gamma bce'),  pt' = (1 - 2 t) p (1 - p),  bce' = p - t -- written
      // the way autograd differentiates max(x, 0) - x t + log1p(exp(-|x|)), so that a logit of exactly 0 gets the reference's sub-gradient
      // (clamp passes the gradient at its bound, |x| has slope 0 at 0: 1 - t there, not 1/2 - t)
      const float dmod = gamma == 2.f ? 2.f * f.pt : (f.pt > 0.f ? gamma * powf(f.pt, gamma - 1.f) : 0.f);
      const float dpt = (1.f - 2.f * tv) * f.p * (1.f - f.p);
      const float e = expf(-fabsf(xv));
      const float dbce = (xv >= 0.f ? 1.f : 0.f) - tv - (xv > 0.f ? 1.f : (xv < 0.f ? -1.f : 0.f)) * (e / (1.f + e));
      out[i] = gout[i] * wv * f.alpha_w * (dmod * dpt * f.bce + mod * dbce);
    }
  }
}

extern "C" int sv_sigmoid_focal_loss(const float* input, const float* target, const float* weights, int64_t n_rows, int num_class, float alpha, float gamma,
                                     const float* grad_out, float* out, void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && num_class > 0, "sigmoid_focal_loss: bad sizes");
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(input && target && out, "sigmoid_focal_loss: null pointer");
  const int64_t total = n_rows * num_class;
  const dim3 grid(sv_grid_1d(total, 256, 256 * 16));
  if (grad_out) hipLaunchKernelGGL(k_sigmoid_focal<true>, grid, dim3(256), 0, sv_stream(stream), total, num_class, input, target, weights, alpha, gamma, grad_out, out);
  else hipLaunchKernelGGL(k_sigmoid_focal<false>, grid, dim3(256), 0, sv_stream(stream), total, num_class, input, target, weights, alpha, gamma, nullptr, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

template <bool GRAD>
__global__ __launch_bounds__(256) void k_weighted_smooth_l1(int64_t total, int C, const float* __restrict__ x, const float* __restrict__ t,
                                                             const float* __restrict__ cw, const float* __restrict__ w, float beta,
                                                             const float* __restrict__ gout, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float xv = x[i], tr = t[i];
    const float tv = isnan(tr) ? xv : tr;                               // nan targets are ignored (loss_utils.py:117)
    const float cwv = cw ? cw[i % C] : 1.f, wv = w ? w[i / C] : 1.f;
    const float d = (xv - tv) * cwv, n = fabsf(d);
    if (!GRAD) {
      const float l = beta < 1e-5f ? n : (n < beta ? 0.5f * (n * n) / beta : n - 0.5f * beta);
      out[i] = l * wv;
    } else {
      const float dl = isnan(tr) ? 0.f : (beta < 1e-5f || n >= beta ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : d / beta);
      out[i] = gout[i] * wv * cwv * dl;
    }
  }
}

extern "C" int sv_weighted_smooth_l1_loss(const float* input, const float* target, const float* code_weights, const float* weights, int64_t n_rows,
                                          int num_codes, float beta, const float* grad_out, float* out, void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && num_codes > 0, "weighted_smooth_l1_loss: bad sizes");
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(input && target && out, "weighted_smooth_l1_loss: null pointer");
  const int64_t total = n_rows * num_codes;
  const dim3 grid(sv_grid_1d(total, 256, 256 * 16));
  if (grad_out) hipLaunchKernelGGL(k_weighted_smooth_l1<true>, grid, dim3(256), 0, sv_stream(stream), total, num_codes, input, target, code_weights, weights, beta, grad_out, out);
  else hipLaunchKernelGGL(k_weighted_smooth_l1<false>, grid, dim3(256), 0, sv_stream(stream), total, num_codes, input, target, code_weights, weights, beta, nullptr, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
