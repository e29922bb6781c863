// RoI-aware point feature pooling (PartA2 head) -- replaces roiaware_pool3d_cuda.forward / backward
// (detector3d/pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:39-312, roiaware_pool3d.cpp:60-170).
// The reference materialises an (N boxes x M points) mask, then ONE THREAD PER BOX walks all M points to fill the voxel lists,
// then pools with one thread per (voxel, channel) striding channels across blocks.  Here one 1024-thread workgroup owns a box:
// ordered compaction of the box's points (ballot + prefix), slot assignment in point order by one wave with the voxel counters
// in LDS, pooling with channels fastest (coalesced feature rows).  Same outputs: pts_idx_of_voxels (slot 0 = count, first-come
// in point order, at most max_pts-1 points), argmax, pooled features.
#include "common.h"

#define RA_THREADS 1024
#define RA_WAVES (RA_THREADS / SV_WAVE)
#define RA_MAX_VOXELS 4096

struct RoiPoolArgs {
  const float* rois;      // (N,7) [x,y,z,dx,dy,dz,heading], centre
  const float* pts;       // (M,3)
  const float* feat;      // (M,C)
  int N, M, C, max_pts, ox, oy, oz, method;   // method 0 max, 1 avg
  int32_t* scratch;       // (N,M): ascending indices of the points inside box n
  int32_t* argmax;        // (N,ox,oy,oz,C)
  int32_t* pts_idx;       // (N,ox,oy,oz,max_pts)
  float* pooled;          // (N,ox,oy,oz,C)
};

// check_pt_in_box3d (:23-36) + the voxel of the point (:57-72); returns -1 outside, else x*oy*oz + y*oz + z
__device__ __forceinline__ int ra_voxel_of(const float* p, const float* b, float cosa, float sina, int ox, int oy, int oz) {
  const float x = p[0], y = p[1], z = p[2];
  const float cz = b[2], dx = b[3], dy = b[4], dz = b[5];
  if (fabsf(z - cz) > dz / 2.0) return -1;
  const float sx = x - b[0], sy = y - b[1];
  const float lx = sx * cosa + sy * (-sina), ly = sx * sina + sy * cosa;
  if (!(fabs(lx) < dx / 2.0 + (double)1e-5f && fabs(ly) < dy / 2.0 + (double)1e-5f)) return -1;
  const float lz = z - cz;
  const float xr = dx / ox, yr = dy / oy, zr = dz / oz;
  unsigned xi = (unsigned)(int)((lx + dx / 2) / xr), yi = (unsigned)(int)((ly + dy / 2) / yr), zi = (unsigned)(int)((lz + dz / 2) / zr);
  xi = min(max(xi, 0u), (unsigned)(ox - 1)), yi = min(max(yi, 0u), (unsigned)(oy - 1)), zi = min(max(zi, 0u), (unsigned)(oz - 1));
  xi &= 0xFF, yi &= 0xFF, zi &= 0xFF;                    // the reference packs the three indices into 8 bits each (:74)
  return (int)(xi * oy * oz + yi * oz + zi);
}

__global__ __launch_bounds__(RA_THREADS) void k_roiaware_pool(RoiPoolArgs a) {
  __shared__ int s_cnt[RA_MAX_VOXELS];
  __shared__ int s_wcnt[RA_WAVES];
  __shared__ int s_base;
  const int box = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* b = a.rois + (size_t)box * 7;
  const float cosa = cosf(-b[6]), sina = sinf(-b[6]);
  const int nvox = a.ox * a.oy * a.oz;
  int32_t* list = a.scratch + (size_t)box * a.M;
  for (int v = tid; v < nvox; v += RA_THREADS) s_cnt[v] = 0;
  if (tid == 0) s_base = 0;
  __syncthreads();
  // 1. ascending list of the points inside this box
  for (int start = 0; start < a.M; start += RA_THREADS) {
    const int i = start + tid;
    const bool in = i < a.M && ra_voxel_of(a.pts + (size_t)i * 3, b, cosa, sina, a.ox, a.oy, a.oz) >= 0;
    const unsigned long long vote = __ballot(in);
    if (lane == 0) s_wcnt[wave] = __popcll(vote);
    __syncthreads();
    int pos = s_base + __popcll(vote & ((1ull << lane) - 1));
    for (int w = 0; w < wave; ++w) pos += s_wcnt[w];
    if (in) list[pos] = i;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < RA_WAVES; ++w) t += s_wcnt[w];
      s_base += t;
    }
    __syncthreads();
  }
  const int n_in = s_base;
  int32_t* vox_list = a.pts_idx + (size_t)box * nvox * a.max_pts;
  // 2. first-come slots in point order: wave 0 walks the list 64 at a time; rank inside the chunk by comparing with earlier lanes
  if (wave == 0) {
    for (int start = 0; start < n_in; start += 64) {
      const int r = start + lane;
      const int pt = r < n_in ? list[r] : -1;
      const int v = pt >= 0 ? ra_voxel_of(a.pts + (size_t)pt * 3, b, cosa, sina, a.ox, a.oy, a.oz) : -1;
      int before = 0, after = 0;
      for (int l = 0; l < 64; ++l) {
        const int ov = __shfl(v, l);
        before += (l < lane && ov == v) ? 1 : 0;
        after += (l > lane && ov == v) ? 1 : 0;
      }
      if (v >= 0) {
        const int slot = s_cnt[v] + before;
        if (slot < a.max_pts - 1) vox_list[(size_t)v * a.max_pts + slot + 1] = pt;
        if (after == 0) s_cnt[v] = slot + 1;             // last point of this voxel in the chunk publishes the new count
      }
      __threadfence_block();
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int v = tid; v < nvox; v += RA_THREADS) vox_list[(size_t)v * a.max_pts] = min(s_cnt[v], a.max_pts - 1);
  // 3. pool, channels fastest
  const int64_t total = (int64_t)nvox * a.C;
  for (int64_t e = tid; e < total; e += RA_THREADS) {
    const int v = (int)(e / a.C), c = (int)(e - (int64_t)v * a.C);
    const int cnt = min(s_cnt[v], a.max_pts - 1);
    const int32_t* l = vox_list + (size_t)v * a.max_pts + 1;
    const size_t o = ((size_t)box * nvox + v) * a.C + c;
    if (a.method == 0) {
      int am = -1;
      float best = -__builtin_inff();                    // the reference's float(-1e50)
      for (int k = 0; k < cnt; ++k) {
        const float f = a.feat[(size_t)l[k] * a.C + c];
        if (f > best) best = f, am = l[k];
      }
      a.pooled[o] = am >= 0 ? best : 0.f;
      a.argmax[o] = am;
    } else {
      float s = 0.f;
      for (int k = 0; k < cnt; ++k) s += a.feat[(size_t)l[k] * a.C + c];
      a.pooled[o] = cnt > 0 ? s / cnt : 0.f;
    }
  }
}

// backward (:232-312): grad_in (M,C) accumulates (caller zero-fills, like the reference's new_zeros)
__global__ __launch_bounds__(256) void k_roiaware_pool_bwd(int64_t total, int C, int max_pts, int method, const int32_t* __restrict__ pts_idx,
                                                           const int32_t* __restrict__ argmax, const float* __restrict__ grad_out,
                                                           float* __restrict__ grad_in) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (box, voxel, channel), channel fastest
  if (e >= total) return;
  const int64_t bv = e / C;
  const int c = (int)(e - bv * C);
  if (method == 0) {
    const int am = argmax[e];
    if (am >= 0) atomicAdd(&grad_in[(size_t)am * C + c], grad_out[e] * 1);
  } else {
    const int32_t* l = pts_idx + (size_t)bv * max_pts;
    const int cnt = l[0];
    const float g = 1 / fmaxf((float)cnt, 1.0f);
    for (int k = 1; k <= cnt; ++k) atomicAdd(&grad_in[(size_t)l[k] * C + c], grad_out[e] * g);
  }
}

// points_in_boxes_cpu (roiaware_pool3d.cpp:121-165): (N boxes x M points) 0/1 matrix, MARGIN 1e-2 (the GPU variant uses 1e-5)
__global__ __launch_bounds__(256) void k_points_in_boxes_matrix(int N, int M, const float* __restrict__ boxes, const float* __restrict__ pts,
                                                                int32_t* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= M) return;
  const float* b = boxes + (size_t)i * 7;
  const float x = pts[(size_t)j * 3], y = pts[(size_t)j * 3 + 1], z = pts[(size_t)j * 3 + 2];
  int in = 0;
  if (!(fabsf(z - b[2]) > b[5] / 2.0)) {
    const float cosa = cosf(-b[6]), sina = sinf(-b[6]);
    const float sx = x - b[0], sy = y - b[1];
    const float lx = sx * cosa + sy * (-sina), ly = sx * sina + sy * cosa;
    in = (fabs(lx) < b[3] / 2.0 + (double)1e-2f && fabs(ly) < b[4] / 2.0 + (double)1e-2f) ? 1 : 0;
  }
  out[(size_t)i * M + j] = in;
}

extern "C" int sv_points_in_boxes_matrix(const float* boxes, const float* pts, int num_boxes, int num_points, int32_t* out, void* stream) {
  SV_CHECK_ARG(num_boxes >= 0 && num_points >= 0, "sv_points_in_boxes_matrix: negative size");
  if (num_boxes == 0 || num_points == 0) return SV_OK;
  SV_CHECK_ARG(boxes && pts && out, "sv_points_in_boxes_matrix: null pointer");
  hipLaunchKernelGGL(k_points_in_boxes_matrix, dim3(sv_div_up(num_points, 256), num_boxes), dim3(256), 0, sv_stream(stream), num_boxes,
                     num_points, boxes, pts, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int64_t sv_roiaware_pool3d_scratch_bytes(int num_rois, int num_pts) { return (int64_t)num_rois * num_pts * 4; }

extern "C" int sv_roiaware_pool3d_forward(const float* rois, const float* pts, const float* pts_feature, int num_rois, int num_pts, int channels,
                                          int out_x, int out_y, int out_z, int max_pts_each_voxel, int pool_method, void* scratch,
                                          int32_t* argmax, int32_t* pts_idx_of_voxels, float* pooled_features, void* stream) {
  SV_CHECK_ARG(num_rois >= 0 && num_pts >= 0 && channels >= 1 && max_pts_each_voxel >= 1, "sv_roiaware_pool3d_forward: bad sizes");
  SV_CHECK_ARG(out_x >= 1 && out_y >= 1 && out_z >= 1 && out_x <= 256 && out_y <= 256 && out_z <= 256 &&
                   (int64_t)out_x * out_y * out_z <= RA_MAX_VOXELS,
               "sv_roiaware_pool3d_forward: at most %d voxels per RoI (got %dx%dx%d)", RA_MAX_VOXELS, out_x, out_y, out_z);
  SV_CHECK_ARG(pool_method == 0 || pool_method == 1, "sv_roiaware_pool3d_forward: pool_method 0 (max) or 1 (avg)");
  if (num_rois == 0) return SV_OK;
  SV_CHECK_ARG(rois && pts_idx_of_voxels && pooled_features && (pool_method == 1 || argmax), "sv_roiaware_pool3d_forward: null output");
  SV_CHECK_ARG(num_pts == 0 || (pts && pts_feature && scratch), "sv_roiaware_pool3d_forward: null input");
  RoiPoolArgs a;
  a.rois = rois, a.pts = pts, a.feat = pts_feature, a.N = num_rois, a.M = num_pts, a.C = channels, a.max_pts = max_pts_each_voxel;
  a.ox = out_x, a.oy = out_y, a.oz = out_z, a.method = pool_method, a.scratch = static_cast<int32_t*>(scratch), a.argmax = argmax;
  a.pts_idx = pts_idx_of_voxels, a.pooled = pooled_features;
  hipLaunchKernelGGL(k_roiaware_pool, dim3(num_rois), dim3(RA_THREADS), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_roiaware_pool3d_backward(const int32_t* pts_idx_of_voxels, const int32_t* argmax, const float* grad_out, int num_rois,
                                           int out_x, int out_y, int out_z, int channels, int max_pts_each_voxel, int pool_method,
                                           float* grad_in, void* stream) {
  SV_CHECK_ARG(num_rois >= 0 && channels >= 1 && (pool_method == 0 || pool_method == 1), "sv_roiaware_pool3d_backward: bad arguments");
  const int64_t total = (int64_t)num_rois * out_x * out_y * out_z * channels;
  if (total == 0) return SV_OK;
  SV_CHECK_ARG(grad_out && grad_in && (pool_method == 0 ? argmax != nullptr : pts_idx_of_voxels != nullptr),
               "sv_roiaware_pool3d_backward: null pointer");
  hipLaunchKernelGGL(k_roiaware_pool_bwd, dim3(sv_div_up(total, 256)), dim3(256), 0, sv_stream(stream), total, channels, max_pts_each_voxel,
                     pool_method, pts_idx_of_voxels, argmax, grad_out, grad_in);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
