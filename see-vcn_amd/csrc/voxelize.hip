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
                                                          VoxGeom g, SvIndexView ix, int64_t* __restrict__ keys, int32_t* __restrict__ cnt,
                                                          int64_t capacity, float* __restrict__ feats, int64_t feat_words) {
  // the accumulators of k_vox_accum (two launches later) start from zero: cleared here instead of by two memsets
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += (int64_t)gridDim.x * blockDim.x) cnt[i] = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < feat_words; i += (int64_t)gridDim.x * blockDim.x) feats[i] = 0.f;
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

  const int grid = sv_grid_1d(num_points, VOX_THREADS);
  if (num_points > 0) {
    hipLaunchKernelGGL(k_vox_mark, dim3(grid), dim3(VOX_THREADS), 0, st, points, num_points, point_stride, g, ix, keys, cnt, capacity, voxel_features,
                       capacity * num_features);
  } else {
    SV_HIP(hipMemsetAsync(cnt, 0, (size_t)capacity * 4, st));
    SV_HIP(hipMemsetAsync(voxel_features, 0, (size_t)capacity * num_features * 4, st));
  }
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

// ------------------------------------------------------------------------------------------------
// Hard voxelisation (first-come semantics of spconv's VoxelGenerator / Point2VoxelCPU3d as called by
// DataProcessor.transform_points_to_voxels, detector3d/pcdet/datasets/processor/data_processor.py:15-60,115-143):
//   walk the points in order; a point whose cell is new opens the next voxel unless max_voxels are open (then it is
//   dropped); a point is stored in its voxel's next slot unless the voxel already holds max_points points.
// The reference runs this on the CPU inside dataloader workers.  Here, over the whole chip (grid = point blocks x scenes):
//   (1) k_hv_insert: cells go into an open-addressing hash table per scene (64-bit key, CAS); every point takes part in an atomicMin
//       of its cell's FIRST point index and pushes itself on its cell's list (atomicExch of the head, no order implied);
//   (2) a point is "first" iff it owns its cell's minimum: k_hv_count / k_hv_offsets / k_hv_fill are an in-order scan of those flags over
//       the points of a scene = the voxel numbers in order of first appearance (no sort), capped at max_voxels;
//   (3) k_hv_fill, the thread of a first point whose voxel is open: walks the cell's list once, keeps the max_points SMALLEST point indices
//       in ascending order (insertion into an LDS column) -- exactly the points the sequential walk would have stored, in its order --
//       and writes the voxel's rows, coordinates and count.
// (Round 1 ran all of this in ONE workgroup per scene: 12 ms for a 300 k-point nuScenes-shaped scene.)
// Algorithmic bytes: 4*(1+C)*P read + (4*mp*C + 12 + 4)*V written.
// ------------------------------------------------------------------------------------------------
constexpr int HV_THREADS = 256;
constexpr unsigned long long HV_EMPTY = 0xFFFFFFFFFFFFFFFFull;

struct HardVoxArgs {
  const float* points;            // (sum P, stride) rows [x,y,z,...] (no batch column) or with batch column skipped by `xyz_offset`
  const int32_t* scene_start;     // (B) first row of each scene
  const int32_t* scene_cnt;       // (B)
  int stride, xyz_offset, C;      // floats per row; column of x; features copied = columns xyz_offset .. xyz_offset+C-1
  VoxGeom g;
  int max_points, max_voxels;
  unsigned long long* tab_key;    // (B, tab_size)   0xFF.. = empty
  int32_t* tab_head;              // (B, tab_size)   last point pushed on the cell's list, -1 = none
  int32_t* tab_first;             // (B, tab_size)   smallest point index of the cell (scene-relative)
  int32_t* pt_slot;               // (sum P) table slot of each point or -1
  int32_t* pt_next;               // (sum P) next point of the same cell (scene-relative index) or -1
  int32_t* block_off;             // (B, blocks_per_scene) first-point count of each point block, then its exclusive scan
  int tab_size;                   // power of two >= 2 * max scene size
  int blocks_per_scene;
  float* voxels;                  // (B, max_voxels, max_points, C)
  int32_t* coords;                // (B, max_voxels, 3) [z,y,x]
  int32_t* num_points;            // (B, max_voxels)
  int32_t* num_voxels;            // (B)
};

__global__ __launch_bounds__(HV_THREADS) void k_hv_insert(HardVoxArgs a) {
  const int b = blockIdx.y, i = blockIdx.x * HV_THREADS + threadIdx.x;
  const int p0 = a.scene_start[b], n = a.scene_cnt[b];
  if (i >= n) return;
  unsigned long long* tkey = a.tab_key + (int64_t)b * a.tab_size;
  int32_t* tfirst = a.tab_first + (int64_t)b * a.tab_size;
  int32_t* thead = a.tab_head + (int64_t)b * a.tab_size;
  const unsigned int tmask = (unsigned int)a.tab_size - 1u;
  const float* p = a.points + (int64_t)(p0 + i) * a.stride + a.xyz_offset;
  const float fx = floorf(__fdiv_rn(__fsub_rn(p[0], a.g.lo[0]), a.g.vs[0]));
  const float fy = floorf(__fdiv_rn(__fsub_rn(p[1], a.g.lo[1]), a.g.vs[1]));
  const float fz = floorf(__fdiv_rn(__fsub_rn(p[2], a.g.lo[2]), a.g.vs[2]));
  int slot = -1, next = -1;
  if (fx >= 0.f && fx < (float)a.g.grid[0] && fy >= 0.f && fy < (float)a.g.grid[1] && fz >= 0.f && fz < (float)a.g.grid[2]) {
    const unsigned long long key = ((unsigned long long)(int)fz * a.g.grid[1] + (int)fy) * a.g.grid[0] + (int)fx;
    unsigned int h = (unsigned int)((key * 0x9E3779B97F4A7C15ull) >> 40) & tmask;
    while (true) {
      const unsigned long long prev = atomicCAS(&tkey[h], HV_EMPTY, key);
      if (prev == HV_EMPTY || prev == key) break;
      h = (h + 1) & tmask;
    }
    slot = (int)h;
    atomicMin(&tfirst[slot], i);
    next = atomicExch(&thead[slot], i);
  }
  a.pt_slot[p0 + i] = slot;
  a.pt_next[p0 + i] = next;
}

__device__ __forceinline__ int hv_block_excl_scan(int v, int* total, int* wsum /* [HV_THREADS / 64] */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < HV_THREADS / 64; ++i) {
    const int s = wsum[i];
    if (i < wid) base += s;
    tot += s;
  }
  *total = tot;
  return base + incl - v;
}

__device__ __forceinline__ bool hv_is_first(const HardVoxArgs& a, int b, int p0, int n, int i) {
  if (i >= n) return false;
  const int s = a.pt_slot[p0 + i];
  return s >= 0 && a.tab_first[(int64_t)b * a.tab_size + s] == i;
}

__global__ __launch_bounds__(HV_THREADS) void k_hv_count(HardVoxArgs a) {
  __shared__ int wsum[HV_THREADS / 64];
  const int b = blockIdx.y, i = blockIdx.x * HV_THREADS + threadIdx.x;
  const int p0 = a.scene_start[b], n = a.scene_cnt[b];
  int total;
  hv_block_excl_scan(hv_is_first(a, b, p0, n, i) ? 1 : 0, &total, wsum);
  if (threadIdx.x == 0) a.block_off[(int64_t)b * a.blocks_per_scene + blockIdx.x] = total;
}

// one workgroup per scene: exclusive scan of the block counts in place, number of open voxels
__global__ __launch_bounds__(1024) void k_hv_offsets(HardVoxArgs a) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int32_t* off = a.block_off + (int64_t)b * a.blocks_per_scene;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < a.blocks_per_scene; base += 1024) {
    const int j = base + tid;
    const int v = j < a.blocks_per_scene ? off[j] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int pre = carry, tot = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < wid) pre += wsum[w];
      tot += wsum[w];
    }
    if (j < a.blocks_per_scene) off[j] = pre + incl - v;
    __syncthreads();
    if (tid == 0) carry += tot;
    __syncthreads();
  }
  if (tid == 0) a.num_voxels[b] = min(carry, a.max_voxels);
}

__global__ __launch_bounds__(HV_THREADS) void k_hv_fill(HardVoxArgs a) {
  extern __shared__ int32_t s_sel[];                             // [max_points][HV_THREADS]: column tid = the cell's smallest indices, ascending
  __shared__ int wsum[HV_THREADS / 64];
  const int b = blockIdx.y, tid = threadIdx.x, i = blockIdx.x * HV_THREADS + tid;
  const int p0 = a.scene_start[b], n = a.scene_cnt[b];
  const bool first = hv_is_first(a, b, p0, n, i);
  int total;
  const int vid = a.block_off[(int64_t)b * a.blocks_per_scene + blockIdx.x] + hv_block_excl_scan(first ? 1 : 0, &total, wsum);
  if (!first || vid >= a.max_voxels) return;
  const int slot = a.pt_slot[p0 + i];
  const int K = a.max_points;
  int len = 0, kept = 0;
  for (int j = a.tab_head[(int64_t)b * a.tab_size + slot]; j >= 0; j = a.pt_next[p0 + j]) {
    ++len;
    if (kept == K && j > s_sel[(K - 1) * HV_THREADS + tid]) continue;
    int q = kept < K ? kept : K - 1;
    while (q > 0 && s_sel[(q - 1) * HV_THREADS + tid] > j) {
      s_sel[q * HV_THREADS + tid] = s_sel[(q - 1) * HV_THREADS + tid];
      --q;
    }
    s_sel[q * HV_THREADS + tid] = j;
    if (kept < K) ++kept;
  }
  const unsigned long long key = a.tab_key[(int64_t)b * a.tab_size + slot];
  int32_t* crd = a.coords + ((int64_t)b * a.max_voxels + vid) * 3;
  crd[2] = (int)(key % a.g.grid[0]);
  crd[1] = (int)((key / a.g.grid[0]) % a.g.grid[1]);
  crd[0] = (int)(key / ((unsigned long long)a.g.grid[0] * a.g.grid[1]));
  a.num_points[(int64_t)b * a.max_voxels + vid] = kept;
  float* dst = a.voxels + ((int64_t)b * a.max_voxels + vid) * K * a.C;
  for (int r = 0; r < kept; ++r) {
    const float* p = a.points + (int64_t)(p0 + s_sel[r * HV_THREADS + tid]) * a.stride + a.xyz_offset;
    for (int c = 0; c < a.C; ++c) dst[r * a.C + c] = p[c];
  }
  (void)len;
}

static int64_t hv_table_size(int max_scene_points) {
  int64_t tab = 64;
  while (tab < 2ll * max_scene_points) tab <<= 1;
  return tab;
}

extern "C" size_t sv_voxelize_hard_scratch_bytes(int batch, int64_t total_points, int max_scene_points) {
  const int64_t tab = hv_table_size(max_scene_points);
  const int64_t blocks = (max_scene_points + HV_THREADS - 1) / HV_THREADS + 1;
  return (size_t)batch * tab * (8 + 4 + 4) + (size_t)(total_points > 0 ? total_points : 1) * 8 + (size_t)batch * blocks * 4 + 256;
}

extern "C" int sv_voxelize_hard(const float* points, int point_stride, int xyz_offset, int num_features, const int32_t* scene_start,
                                const int32_t* scene_cnt, int batch, int64_t total_points, int max_scene_points,
                                const float* pc_range_host, const float* voxel_size_host, const int32_t* grid_size_host, int max_points,
                                int max_voxels, void* scratch, float* voxels, int32_t* coords, int32_t* num_points_per_voxel,
                                int32_t* num_voxels, void* stream) {
  SV_CHECK_ARG(batch >= 0 && max_points >= 1 && max_voxels >= 1 && num_features >= 3 && xyz_offset >= 0 &&
                   xyz_offset + num_features <= point_stride, "voxelize_hard: bad arguments");
  if (batch == 0) return SV_OK;
  SV_CHECK_ARG(scene_start && scene_cnt && scratch && voxels && coords && num_points_per_voxel && num_voxels && (total_points == 0 || points),
               "voxelize_hard: null pointer");
  SV_CHECK_ARG((size_t)max_points * HV_THREADS * 4 <= 64 * 1024, "voxelize_hard: at most %d points per voxel", 64 * 1024 / (HV_THREADS * 4));
  hipStream_t st = sv_stream(stream);
  HardVoxArgs a;
  a.points = points; a.scene_start = scene_start; a.scene_cnt = scene_cnt;
  a.stride = point_stride; a.xyz_offset = xyz_offset; a.C = num_features;
  for (int i = 0; i < 3; ++i) {
    a.g.lo[i] = pc_range_host[i]; a.g.vs[i] = voxel_size_host[i]; a.g.grid[i] = grid_size_host[i];
    SV_CHECK_ARG(a.g.grid[i] > 0 && a.g.vs[i] > 0.f, "voxelize_hard: bad grid/voxel size");
  }
  a.g.batch = batch;
  a.max_points = max_points; a.max_voxels = max_voxels;
  const int64_t tab = hv_table_size(max_scene_points);
  a.tab_size = (int)tab;
  a.blocks_per_scene = (max_scene_points + HV_THREADS - 1) / HV_THREADS;
  if (a.blocks_per_scene < 1) a.blocks_per_scene = 1;
  char* s = reinterpret_cast<char*>(scratch);
  a.tab_key = reinterpret_cast<unsigned long long*>(s); s += (size_t)batch * tab * 8;
  a.tab_head = reinterpret_cast<int32_t*>(s); s += (size_t)batch * tab * 4;
  a.tab_first = reinterpret_cast<int32_t*>(s); s += (size_t)batch * tab * 4;
  const size_t np = (size_t)(total_points > 0 ? total_points : 1);
  a.pt_slot = reinterpret_cast<int32_t*>(s); s += np * 4;
  a.pt_next = reinterpret_cast<int32_t*>(s); s += np * 4;
  a.block_off = reinterpret_cast<int32_t*>(s);
  a.voxels = voxels; a.coords = coords; a.num_points = num_points_per_voxel; a.num_voxels = num_voxels;
  // empty keys and list heads are all-ones, the running minimum starts at 0x7F7F7F7F (above any point index); padded outputs are zeros
  SV_HIP(hipMemsetAsync(a.tab_key, 0xFF, (size_t)batch * tab * 12, st));
  SV_HIP(hipMemsetAsync(a.tab_first, 0x7F, (size_t)batch * tab * 4, st));
  SV_HIP(hipMemsetAsync(voxels, 0, (size_t)batch * max_voxels * max_points * num_features * 4, st));
  SV_HIP(hipMemsetAsync(coords, 0, (size_t)batch * max_voxels * 3 * 4, st));
  SV_HIP(hipMemsetAsync(num_points_per_voxel, 0, (size_t)batch * max_voxels * 4, st));
  dim3 grid(a.blocks_per_scene, batch);
  hipLaunchKernelGGL(k_hv_insert, grid, dim3(HV_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_hv_count, grid, dim3(HV_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_hv_offsets, dim3(batch), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(k_hv_fill, grid, dim3(HV_THREADS), (size_t)max_points * HV_THREADS * 4, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// PillarVFE feature decoration (backbones_3d/vfe/pillar_vfe.py:94-118): per point of a pillar
//   [raw C, xyz - mean(xyz of the pillar's points), xyz - pillar centre] (+ optional range), padded slots zeroed.
// voxels (V, mp, C), num_points (V), coords (V,4) [b,z,y,x] -> out (V, mp, C+6(+1))
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pillar_decorate(const float* __restrict__ voxels, const int32_t* __restrict__ nump,
                                                         const int32_t* __restrict__ coords, int64_t V, int mp, int C, float vx, float vy, float vz,
                                                         float ox, float oy, float oz, int use_abs_xyz, int with_distance, float* __restrict__ out) {
  const int Cin = use_abs_xyz ? C : C - 3;
  const int Co = Cin + 6 + (with_distance ? 1 : 0);
  const int64_t total = V * mp;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = e / mp;
    const int s = (int)(e - v * mp);
    const float* base = voxels + v * mp * C;
    const int np = nump[v];
    float mx = 0.f, my = 0.f, mz = 0.f;
    for (int k = 0; k < mp; ++k) { mx += base[k * C]; my += base[k * C + 1]; mz += base[k * C + 2]; }   // sum over ALL slots (:97)
    const float fn = (float)np;
    mx = __fdiv_rn(mx, fn); my = __fdiv_rn(my, fn); mz = __fdiv_rn(mz, fn);
    const float* p = base + s * C;
    const float m = s < np ? 1.f : 0.f;
    float* o = out + e * Co;
    int w = 0;
    for (int c = use_abs_xyz ? 0 : 3; c < C; ++c) o[w++] = p[c] * m;
    o[w++] = (p[0] - mx) * m; o[w++] = (p[1] - my) * m; o[w++] = (p[2] - mz) * m;
    const int4 cd = reinterpret_cast<const int4*>(coords)[v];
    o[w++] = (p[0] - ((float)cd.w * vx + ox)) * m;
    o[w++] = (p[1] - ((float)cd.z * vy + oy)) * m;
    o[w++] = (p[2] - ((float)cd.y * vz + oz)) * m;
    if (with_distance) o[w++] = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) * m;
  }
}

extern "C" int sv_pillar_decorate(const float* voxels, const int32_t* num_points, const int32_t* coords, int64_t num_voxels, int max_points,
                                  int num_features, const float* voxel_size_host, const float* pc_range_host, int use_absolute_xyz,
                                  int with_distance, float* out, void* stream) {
  SV_CHECK_ARG(num_voxels >= 0 && max_points >= 1 && num_features >= 3, "pillar_decorate: bad arguments");
  if (num_voxels == 0) return SV_OK;
  SV_CHECK_ARG(voxels && num_points && coords && out, "pillar_decorate: null pointer");
  const float vx = voxel_size_host[0], vy = voxel_size_host[1], vz = voxel_size_host[2];
  // offsets are computed in double by the reference's python (voxel/2 + range) and used as python floats
  const float ox = (float)((double)vx / 2 + pc_range_host[0]), oy = (float)((double)vy / 2 + pc_range_host[1]),
              oz = (float)((double)vz / 2 + pc_range_host[2]);
  hipLaunchKernelGGL(k_pillar_decorate, dim3(sv_grid_1d(num_voxels * max_points, 256)), dim3(256), 0, sv_stream(stream), voxels, num_points,
                     coords, num_voxels, max_points, num_features, vx, vy, vz, ox, oy, oz, use_absolute_xyz, with_distance, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
