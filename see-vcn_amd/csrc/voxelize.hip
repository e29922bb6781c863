// Dynamic voxelisation + mean VFE (HBM-bound; algorithmic bytes = 16*P + 28*V for xyz-only points).
//
// No sort: points set bits in the persistent coordinate index (dense bitmap in HBM), a scan over the
// per-chunk counts gives every occupied cell its position in ascending key order (= torch.unique order
// of dynamic_mean_vfe.py:62), points then accumulate into their voxel row with fp32 atomics and the last
// pass divides by the count and returns the touched index words to zero.
#include "common.h"

struct VoxGeom {
  float lo[3];
  float vs[3];
  int grid[3];  // X, Y, Z
  int batch;
};

constexpr int VOX_THREADS = 256;

__device__ __forceinline__ int64_t vox_key(const float* __restrict__ p, const VoxGeom& g) {
  // floor((xyz - min) / voxel) exactly as torch does it in fp32 (dynamic_mean_vfe.py:53): IEEE sub, IEEE div
  const float fx = floorf(__fdiv_rn(__fsub_rn(p[1], g.lo[0]), g.vs[0]));
  const float fy = floorf(__fdiv_rn(__fsub_rn(p[2], g.lo[1]), g.vs[1]));
  const float fz = floorf(__fdiv_rn(__fsub_rn(p[3], g.lo[2]), g.vs[2]));
  const int b = (int)p[0];
  const bool ok = fx >= 0.f && fx < (float)g.grid[0] && fy >= 0.f && fy < (float)g.grid[1] && fz >= 0.f &&
                  fz < (float)g.grid[2] && b >= 0 && b < g.batch;
  if (!ok) return -1;
  // key = b*XYZ + x*YZ + y*Z + z  (dynamic_mean_vfe.py:57-60), in 64 bits
  return (((int64_t)b * g.grid[0] + (int)fx) * g.grid[1] + (int)fy) * g.grid[2] + (int)fz;
}

__global__ __launch_bounds__(VOX_THREADS) void k_vox_mark(const float* __restrict__ points, int64_t n, int stride,
                                                          VoxGeom g, SvIndexView ix, int64_t* __restrict__ keys) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float p[4];
    if (stride == 4) {
      const float4 q = reinterpret_cast<const float4*>(points)[i];
      p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
    } else {
      const float* r = points + i * stride;
      p[0] = r[0]; p[1] = r[1]; p[2] = r[2]; p[3] = r[3];
    }
    const int64_t key = vox_key(p, g);
    keys[i] = key;
    if (key >= 0) sv_index_mark(ix, key);
  }
}

template <int C>
__global__ __launch_bounds__(VOX_THREADS) void k_vox_accum(const float* __restrict__ points, int64_t n, int stride,
                                                           int nfeat, VoxGeom g, SvIndexView ix,
                                                           const int64_t* __restrict__ keys, int64_t capacity,
                                                           int32_t* __restrict__ coords, float* __restrict__ feats,
                                                           int32_t* __restrict__ cnt, int32_t* __restrict__ p2v) {
  const int nf = C > 0 ? C : nfeat;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t key = keys[i];
    int32_t row = -1;
    if (key >= 0) {
      row = sv_index_rank_slow(ix, key, nullptr);
      if (row < capacity) {
        const float* r = points + i * stride + 1;
        float* dst = feats + (int64_t)row * nf;
#pragma unroll
        for (int c = 0; c < nf; ++c) atomicAdd(dst + c, r[c]);
        if (atomicAdd(cnt + row, 1) == 0) {
          // decode (dynamic_mean_vfe.py:67-71) and reorder to [b, z, y, x]
          const int z = (int)(key % g.grid[2]);
          const int64_t t = key / g.grid[2];
          const int y = (int)(t % g.grid[1]);
          const int64_t u = t / g.grid[1];
          const int x = (int)(u % g.grid[0]);
          const int b = (int)(u / g.grid[0]);
          reinterpret_cast<int4*>(coords)[row] = make_int4(b, z, y, x);
        }
      } else {
        row = -1;
      }
    }
    if (p2v) p2v[i] = row;
  }
}

__global__ __launch_bounds__(VOX_THREADS) void k_vox_finalize(int64_t n, SvIndexView ix,
                                                              const int64_t* __restrict__ keys, int nfeat,
                                                              int64_t capacity, float* __restrict__ feats,
                                                              const int32_t* __restrict__ cnt,
                                                              int32_t* __restrict__ num_voxels) {
  int64_t nv = *num_voxels;
  if (nv > capacity) nv = capacity;
  const int64_t work = n > nv ? n : nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < work; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nv) {
      const float c = (float)cnt[i];  // >= 1 by construction
      float* dst = feats + i * nfeat;
      for (int k = 0; k < nfeat; ++k) dst[k] = __fdiv_rn(dst[k], c);
    }
    if (i < n) {
      const int64_t key = keys[i];
      if (key >= 0) {  // return the index to all-zero (idempotent across duplicate keys)
        ix.words[key >> 5] = make_uint2(0u, 0u);
        ix.chunk_cnt[key >> SV_CHUNK_SHIFT] = 0;
      }
    }
  }
  // clamp the reported count so callers can trust num_voxels <= capacity
  if (blockIdx.x == 0 && threadIdx.x == 0 && *num_voxels > capacity) *num_voxels = (int32_t)capacity;
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

extern "C" size_t sv_voxelize_dynamic_scratch_bytes(int64_t num_points, int64_t ncells, int64_t capacity) {
  return align256((size_t)num_points * 8) + align256((size_t)capacity * 4) + align256(sv_index_scan_tmp_bytes(ncells));
}

extern "C" int sv_voxelize_dynamic(const float* points, int64_t num_points, int point_stride, int num_features,
                                   const float* pc_range_host, const float* voxel_size_host,
                                   const int32_t* grid_size_host, int batch_size, void* index_ws, void* scratch,
                                   int32_t* voxel_coords, float* voxel_features, int32_t* point_to_voxel,
                                   int64_t capacity, int32_t* num_voxels, void* stream) {
  SV_CHECK_ARG(point_stride >= 4, "voxelize_dynamic: point_stride %d < 4 (need [b,x,y,z,...])", point_stride);
  SV_CHECK_ARG(num_features >= 1 && num_features <= point_stride - 1, "voxelize_dynamic: num_features %d out of range",
               num_features);
  SV_CHECK_ARG(batch_size >= 1 && num_points >= 0 && capacity >= 0, "voxelize_dynamic: bad sizes");
  SV_CHECK_ARG(index_ws && scratch && voxel_coords && voxel_features && num_voxels, "voxelize_dynamic: null pointer");
  SV_CHECK_ARG(num_points == 0 || points, "voxelize_dynamic: null points");
  hipStream_t st = sv_stream(stream);
  VoxGeom g;
  for (int i = 0; i < 3; ++i) {
    g.lo[i] = pc_range_host[i];
    g.vs[i] = voxel_size_host[i];
    g.grid[i] = grid_size_host[i];
    SV_CHECK_ARG(g.grid[i] > 0 && g.vs[i] > 0.f, "voxelize_dynamic: bad grid/voxel size");
  }
  g.batch = batch_size;
  const int64_t ncells = (int64_t)batch_size * g.grid[0] * g.grid[1] * g.grid[2];
  SvIndexView ix = sv_index_view(index_ws, ncells);

  char* s = reinterpret_cast<char*>(scratch);
  int64_t* keys = reinterpret_cast<int64_t*>(s);
  s += align256((size_t)num_points * 8);
  int32_t* cnt = reinterpret_cast<int32_t*>(s);
  s += align256((size_t)capacity * 4);
  void* scan_tmp = s;

  SV_HIP(hipMemsetAsync(cnt, 0, (size_t)capacity * 4, st));
  SV_HIP(hipMemsetAsync(voxel_features, 0, (size_t)capacity * num_features * 4, st));
  const int grid = sv_grid_1d(num_points, VOX_THREADS);
  if (num_points > 0)
    hipLaunchKernelGGL(k_vox_mark, dim3(grid), dim3(VOX_THREADS), 0, st, points, num_points, point_stride, g, ix, keys);
  int rc = sv_index_scan_launch(ix, num_voxels, scan_tmp, st);
  if (rc) return rc;
  if (num_points > 0) {
    if (num_features == 3)
      hipLaunchKernelGGL(k_vox_accum<3>, dim3(grid), dim3(VOX_THREADS), 0, st, points, num_points, point_stride,
                         num_features, g, ix, keys, capacity, voxel_coords, voxel_features, cnt, point_to_voxel);
    else if (num_features == 4)
      hipLaunchKernelGGL(k_vox_accum<4>, dim3(grid), dim3(VOX_THREADS), 0, st, points, num_points, point_stride,
                         num_features, g, ix, keys, capacity, voxel_coords, voxel_features, cnt, point_to_voxel);
    else
      hipLaunchKernelGGL(k_vox_accum<0>, dim3(grid), dim3(VOX_THREADS), 0, st, points, num_points, point_stride,
                         num_features, g, ix, keys, capacity, voxel_coords, voxel_features, cnt, point_to_voxel);
  }
  const int64_t work = num_points > capacity ? num_points : capacity;
  hipLaunchKernelGGL(k_vox_finalize, dim3(sv_grid_1d(work, VOX_THREADS)), dim3(VOX_THREADS), 0, st, num_points, ix,
                     keys, num_features, capacity, voxel_features, cnt, num_voxels);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// MeanVFE (mean_vfe.py:25-29): one thread per (voxel, channel) would waste lanes at C=3; one thread per
// voxel reads max_points*C contiguous floats (60 B at 5x3) — HBM-bound, 4*(mp*C + 1 + C) bytes/voxel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mean_vfe(const float* __restrict__ voxels, const int32_t* __restrict__ nump,
                                                  int64_t nv, int mp, int C, float* __restrict__ out) {
  const int64_t total = nv * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / C;
    const int c = (int)(i - v * C);
    const float* src = voxels + v * mp * C + c;
    float s = 0.f;
    for (int k = 0; k < mp; ++k) s = __fadd_rn(s, src[(int64_t)k * C]);  // torch sums the full point axis in order
    const float d = fmaxf((float)nump[v], 1.0f);
    out[i] = __fdiv_rn(s, d);
  }
}

extern "C" int sv_mean_vfe(const float* voxels, const int32_t* num_points, int64_t num_voxels, int max_points,
                           int num_features, float* out, void* stream) {
  SV_CHECK_ARG(num_voxels >= 0 && max_points >= 1 && num_features >= 1, "mean_vfe: bad sizes");
  if (num_voxels == 0) return SV_OK;
  SV_CHECK_ARG(voxels && num_points && out, "mean_vfe: null pointer");
  hipLaunchKernelGGL(k_mean_vfe, dim3(sv_grid_1d(num_voxels * num_features, 256)), dim3(256), 0, sv_stream(stream),
                     voxels, num_points, num_voxels, max_points, num_features, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
