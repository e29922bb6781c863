// Sparse -> dense BEV scatter (HBM-write bound: 4*B*C*D*H*W bytes written once + 4*V*C read).
//
// SparseConvTensor.dense() as used by HeightCompression (detector3d/pcdet/models/backbones_2d/map_to_bev/
// height_compression.py:21-23) and PointPillarScatter (pointpillar_scatter.py:14-37).  Instead of a memset
// followed by a scatter (every output byte written twice on the occupied part and the zero fill as a separate
// launch), a small cell->row map is scattered first and ONE pass writes every output element exactly once,
// coalesced along x.
#include "common.h"

constexpr int DN_THREADS = 256;

__global__ __launch_bounds__(DN_THREADS) void k_cellmap_scatter(const int4* __restrict__ coords, int64_t n, int B, int D, int H, int W,
                                                                int32_t* __restrict__ cellmap) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = coords[i];
    if (c.x >= 0 && c.x < B && c.y >= 0 && c.y < D && c.z >= 0 && c.z < H && c.w >= 0 && c.w < W)
      cellmap[(((int64_t)c.x * D + c.y) * H + c.z) * W + c.w] = (int32_t)i;
  }
}

// out (B, C, D, H, W); one thread per spatial cell, looping over channels (writes coalesced over x per channel)
__global__ __launch_bounds__(DN_THREADS) void k_dense_write(const float* __restrict__ feat, const int32_t* __restrict__ cellmap, int B, int C,
                                                            int64_t spatial /*D*H*W*/, float* __restrict__ out) {
  const int64_t total = (int64_t)B * spatial;
  for (int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < total; cell += (int64_t)gridDim.x * blockDim.x) {
    const int32_t row = cellmap[cell];
    const int64_t b = cell / spatial, s = cell - b * spatial;
    float* o = out + (b * C) * spatial + s;
    if (row < 0) {
      for (int c = 0; c < C; ++c) o[(int64_t)c * spatial] = 0.f;
    } else {
      const float* f = feat + (int64_t)row * C;
      for (int c = 0; c < C; c += 4) {
        if (c + 4 <= C) {
          const float4 v = *reinterpret_cast<const float4*>(f + c);
          o[(int64_t)c * spatial] = v.x; o[(int64_t)(c + 1) * spatial] = v.y;
          o[(int64_t)(c + 2) * spatial] = v.z; o[(int64_t)(c + 3) * spatial] = v.w;
        } else {
          for (int u = c; u < C; ++u) o[(int64_t)u * spatial] = f[u];
        }
      }
    }
  }
}

// 4 consecutive cells per thread, one 16-byte store per channel plane and thread (the tensor is > 99 % zeros: the kernel is a
// 577 MB streaming write at the bench geometry; 4-byte stores reached 3.65 TB/s).  Needs spatial % 4 == 0.
__global__ __launch_bounds__(DN_THREADS) void k_dense_write4(const float* __restrict__ feat, const int32_t* __restrict__ cellmap, int B, int C,
                                                             int64_t spatial, float* __restrict__ out) {
  const int64_t total4 = (int64_t)B * spatial / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = q * 4;
    const int4 rows = *reinterpret_cast<const int4*>(cellmap + cell);
    const int64_t b = cell / spatial, s = cell - b * spatial;
    float* o = out + (b * C) * spatial + s;
    if ((rows.x & rows.y & rows.z & rows.w) < 0 && rows.x < 0 && rows.y < 0 && rows.z < 0 && rows.w < 0) {
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < C; ++c) __builtin_nontemporal_store(z, reinterpret_cast<f32x4_t*>(o + (int64_t)c * spatial));
    } else {
      const float* f0 = rows.x >= 0 ? feat + (int64_t)rows.x * C : nullptr;
      const float* f1 = rows.y >= 0 ? feat + (int64_t)rows.y * C : nullptr;
      const float* f2 = rows.z >= 0 ? feat + (int64_t)rows.z * C : nullptr;
      const float* f3 = rows.w >= 0 ? feat + (int64_t)rows.w * C : nullptr;
      for (int c = 0; c < C; ++c)
        *reinterpret_cast<float4*>(o + (int64_t)c * spatial) = make_float4(f0 ? f0[c] : 0.f, f1 ? f1[c] : 0.f, f2 ? f2[c] : 0.f, f3 ? f3[c] : 0.f);
    }
  }
}

// 64 consecutive cells per workgroup.  The rows of the occupied cells are read once, coalesced, into LDS; then wave w writes the
// channel planes c = w, w+4, ...: 64 lanes x 4 bytes = two full cache lines per store, zeros and values alike.  No lane ever walks
// a feature row with a 4-byte stride (k_dense_write4's slow path, which its whole wave waits for).  spatial % 64 == 0.
constexpr int DT_CELLS = 64;
__global__ __launch_bounds__(DN_THREADS) void k_dense_write_t(const float* __restrict__ feat, const int32_t* __restrict__ cellmap, int C,
                                                              int64_t spatial, float* __restrict__ out) {
  extern __shared__ float s_f[];                       // [slot][C + 1]
  __shared__ int s_slot[DT_CELLS];
  __shared__ int s_row[DT_CELLS];
  __shared__ int s_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t cell0 = (int64_t)blockIdx.x * DT_CELLS;
  const int64_t b = cell0 / spatial, s0 = cell0 - b * spatial;
  if (wave == 0) {
    const int32_t row = cellmap[cell0 + lane];
    const unsigned long long occ = __ballot(row >= 0);
    const int slot = __popcll(occ & ((1ull << lane) - 1));
    s_slot[lane] = row >= 0 ? slot : -1;
    if (row >= 0) s_row[slot] = row;
    if (lane == 0) s_n = __popcll(occ);
  }
  __syncthreads();
  const int n_occ = s_n;
  const int pitch = C + 1;
  // rows -> LDS: half a wave per row, float4 per lane
  for (int slot = tid >> 5; slot < n_occ; slot += DN_THREADS / 32) {
    const float* f = feat + (int64_t)s_row[slot] * C;
    for (int c = (tid & 31) * 4; c < C; c += 128) {
      const float4 v = *reinterpret_cast<const float4*>(f + c);
      float* d = s_f + slot * pitch + c;
      d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    }
  }
  if (n_occ) __syncthreads();
  const int slot = s_slot[lane];
  float* o = out + (b * C) * spatial + s0 + lane;
  for (int c = wave; c < C; c += DN_THREADS / 64) __builtin_nontemporal_store(slot >= 0 ? s_f[slot * pitch + c] : 0.f, o + (int64_t)c * spatial);
}

// backward twin: 64 rows per workgroup; reads walk one channel plane at a time (lanes = consecutive sorted cells), the rows leave
// through LDS as whole 4*C-byte segments instead of 4-byte writes with a 4*C-byte stride.
__global__ __launch_bounds__(DN_THREADS) void k_dense_gather_t(const float* __restrict__ dense, const int4* __restrict__ coords, int64_t n, int B,
                                                               int C, int D, int H, int W, float* __restrict__ out) {
  extern __shared__ float s_f[];                       // [row][C + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t spatial = (int64_t)D * H * W;
  const int64_t i0 = (int64_t)blockIdx.x * 64, i = i0 + lane;
  const int pitch = C + 1;
  int64_t base = -1;
  if (i < n) {
    const int4 p = coords[i];
    if (p.x >= 0 && p.x < B && p.y >= 0 && p.y < D && p.z >= 0 && p.z < H && p.w >= 0 && p.w < W)
      base = (int64_t)p.x * C * spatial + ((int64_t)p.y * H + p.z) * W + p.w;
  }
  for (int c = wave; c < C; c += DN_THREADS / 64) s_f[lane * pitch + c] = base >= 0 ? dense[base + (int64_t)c * spatial] : 0.f;
  __syncthreads();
  const int rows = (int)((n - i0) < 64 ? (n - i0) : 64);
  for (int e = tid; e < rows * C; e += DN_THREADS) {
    const int r = e / C, c = e - r * C;
    out[(i0 + r) * C + c] = s_f[r * pitch + c];
  }
}

extern "C" size_t sv_sparse_to_dense_scratch_bytes(int batch, int D, int H, int W) { return (size_t)batch * D * H * W * sizeof(int32_t); }

extern "C" int sv_sparse_to_dense(const float* features, const int32_t* coords, int64_t n, int batch, int C, int D, int H, int W,
                                  void* scratch, float* out, void* stream) {
  SV_CHECK_ARG(n >= 0 && batch > 0 && C > 0 && D > 0 && H > 0 && W > 0 && out && scratch, "sparse_to_dense: bad arguments");
  SV_CHECK_ARG(n == 0 || (features && coords), "sparse_to_dense: null pointer");
  SV_CHECK_ARG(C % 4 != 0 || ((uintptr_t)features % 16 == 0), "sparse_to_dense: features must be 16-byte aligned");
  hipStream_t st = sv_stream(stream);
  const int64_t spatial = (int64_t)D * H * W;
  int32_t* cellmap = reinterpret_cast<int32_t*>(scratch);
  SV_HIP(hipMemsetAsync(cellmap, 0xFF, (size_t)batch * spatial * 4, st));
  if (n > 0)
    hipLaunchKernelGGL(k_cellmap_scatter, dim3(sv_grid_1d(n, DN_THREADS)), dim3(DN_THREADS), 0, st, reinterpret_cast<const int4*>(coords), n,
                       batch, D, H, W, cellmap);
  const size_t lds_t = (size_t)DT_CELLS * (C + 1) * sizeof(float);
  if (spatial % DT_CELLS == 0 && C % 4 == 0 && lds_t <= 60 * 1024)
    hipLaunchKernelGGL(k_dense_write_t, dim3((unsigned)(batch * spatial / DT_CELLS)), dim3(DN_THREADS), lds_t, st, features, cellmap, C, spatial,
                       out);
  else if (spatial % 4 == 0 && ((uintptr_t)out % 16 == 0) && ((uintptr_t)cellmap % 16 == 0))
    hipLaunchKernelGGL(k_dense_write4, dim3(sv_grid_1d(batch * spatial / 4, DN_THREADS, 256 * 16)), dim3(DN_THREADS), 0, st, features, cellmap,
                       batch, C, spatial, out);
  else
    hipLaunchKernelGGL(k_dense_write, dim3(sv_grid_1d(batch * spatial, DN_THREADS, 256 * 16)), dim3(DN_THREADS), 0, st, features, cellmap, batch,
                       C, spatial, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// backward of dense(): grad_features[i][c] = grad_dense[b][c][z][y][x]
__global__ __launch_bounds__(DN_THREADS) void k_dense_gather(const float* __restrict__ dense, const int4* __restrict__ coords, int64_t n, int B,
                                                             int C, int D, int H, int W, float* __restrict__ out) {
  const int64_t spatial = (int64_t)D * H * W;
  const int64_t total = n * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx / n);           // channel-major so that lanes walk neighbouring cells of one channel plane
    const int64_t i = idx - (int64_t)c * n;
    const int4 p = coords[i];
    float v = 0.f;
    if (p.x >= 0 && p.x < B && p.y >= 0 && p.y < D && p.z >= 0 && p.z < H && p.w >= 0 && p.w < W)
      v = dense[((int64_t)p.x * C + c) * spatial + ((int64_t)p.y * H + p.z) * W + p.w];
    out[i * C + c] = v;
  }
}

extern "C" int sv_dense_to_sparse(const float* dense, const int32_t* coords, int64_t n, int batch, int C, int D, int H, int W, float* out,
                                  void* stream) {
  SV_CHECK_ARG(n >= 0 && batch > 0 && C > 0 && D > 0 && H > 0 && W > 0, "dense_to_sparse: bad arguments");
  if (n == 0) return SV_OK;
  SV_CHECK_ARG(dense && coords && out, "dense_to_sparse: null pointer");
  const size_t lds_t = (size_t)64 * (C + 1) * sizeof(float);
  if (lds_t <= 60 * 1024)
    hipLaunchKernelGGL(k_dense_gather_t, dim3((unsigned)sv_div_up(n, 64)), dim3(DN_THREADS), lds_t, sv_stream(stream), dense,
                       reinterpret_cast<const int4*>(coords), n, batch, C, D, H, W, out);
  else
    hipLaunchKernelGGL(k_dense_gather, dim3(sv_grid_1d(n * C, DN_THREADS, 256 * 16)), dim3(DN_THREADS), 0, sv_stream(stream), dense,
                       reinterpret_cast<const int4*>(coords), n, batch, C, D, H, W, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}


// ------------------------------------------------------------------------------------------------
// The same dense volume in CHANNELS-LAST memory, as the 2-D backbone behind HeightCompression wants it (round 6).  `.dense()` gives (B, C, D, H, W),
// HeightCompression views it as (B, C D, H, W) -- channel c D + d -- and BaseBEVBackbone runs channels_last from 8 scenes per batch on, i.e. it first
// copied the whole tensor into (B, H, W, C D) order: 706 + 219 us forward and 1 171 + 203 us backward of pure layout change per SECOND train step
// (577 MB tensor; profiles/r05_d_second_step_sequence.txt).  Here the volume is WRITTEN in that order -- out[b][y][x][c D + d], a (B, C D, H, W)
// tensor with channels_last strides on the torch side -- and its gradient is read in that order.  One workgroup per 16 consecutive (y, x) positions of
// a scene: the D x 16 cell-map entries, the occupied rows into LDS, then every thread stores 16-byte pieces of the positions' 4 C D-byte pixel rows.
constexpr int DH_POS = 16;
extern "C" int sv_sparse_to_dense_nhwc_applies(int C, int D, int H, int W);
__global__ __launch_bounds__(DN_THREADS) void k_dense_write_nhwc(const float* __restrict__ feat, const int32_t* __restrict__ cellmap, int C, int D,
                                                                 int64_t hw, float* __restrict__ out) {
  extern __shared__ float s_f[];                       // [slot][C + 1]
  __shared__ int s_slot[DH_POS * 8];                   // [d][position]: slot of the cell's row in s_f, or -1 (D <= 8)
  __shared__ int s_row[DH_POS * 8];
  __shared__ int s_n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t pos0 = (int64_t)blockIdx.x * DH_POS;   // (b, y, x) flattened: b * hw + y * W + x
  const int64_t b = pos0 / hw, s0 = pos0 - b * hw;
  const int cells = DH_POS * D;
  if (tid < 64) {
    // cells of the workgroup in (d, position) order; at most 128 of them: two sweeps of the first wave
    int n = 0;
    for (int base = 0; base < cells; base += 64) {
      const int e = base + lane;
      int32_t row = -1;
      if (e < cells) row = cellmap[(b * D + e / DH_POS) * hw + s0 + e % DH_POS];
      const unsigned long long occ = __ballot(row >= 0);
      const int slot = n + __popcll(occ & ((1ull << lane) - 1));
      if (e < cells) s_slot[e] = row >= 0 ? slot : -1;
      if (row >= 0) s_row[slot] = row;
      n += __popcll(occ);
    }
    if (lane == 0) s_n = n;
  }
  __syncthreads();
  const int n_occ = s_n, pitch = C + 1;
  for (int slot = tid >> 5; slot < n_occ; slot += DN_THREADS / 32) {
    const float* f = feat + (int64_t)s_row[slot] * C;
    for (int c = (tid & 31) * 4; c < C; c += 128) {
      const float4 v = *reinterpret_cast<const float4*>(f + c);
      float* d = s_f + slot * pitch + c;
      d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    }
  }
  if (n_occ) __syncthreads();
  const int CD = C * D, q_per_pos = CD / 4;             // 16-byte pieces of a position's pixel row
  float* o = out + pos0 * CD;
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  for (int e = tid; e < DH_POS * q_per_pos; e += DN_THREADS) {
    const int p = e / q_per_pos, ch0 = (e - p * q_per_pos) * 4;
    f32x4_t v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = ch0 + j, c = ch / D, d = ch - c * D;
      const int slot = s_slot[d * DH_POS + p];
      v[j] = slot >= 0 ? s_f[slot * pitch + c] : 0.f;
    }
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(o + (int64_t)e * 4));
  }
}

extern "C" int sv_sparse_to_dense_nhwc(const float* features, const int32_t* coords, int64_t n, int batch, int C, int D, int H, int W, void* scratch,
                                       float* out, void* stream) {
  SV_CHECK_ARG(n >= 0 && batch > 0 && C > 0 && D > 0 && H > 0 && W > 0 && out && scratch, "sparse_to_dense_nhwc: bad arguments");
  SV_CHECK_ARG(n == 0 || (features && coords), "sparse_to_dense_nhwc: null pointer");
  const int64_t hw = (int64_t)H * W;
  SV_CHECK_ARG(sv_sparse_to_dense_nhwc_applies(C, D, H, W), "sparse_to_dense_nhwc: C %% 4 == 0, D <= 8, H W %% 16 == 0 and a workgroup's rows in 60 KB of LDS "
               "(ask sv_sparse_to_dense_nhwc_applies; got C %d, D %d, H %d, W %d)", C, D, H, W);
  SV_CHECK_ARG((uintptr_t)features % 16 == 0 && (uintptr_t)out % 16 == 0, "sparse_to_dense_nhwc: 16-byte alignment");
  hipStream_t st = sv_stream(stream);
  int32_t* cellmap = reinterpret_cast<int32_t*>(scratch);
  SV_HIP(hipMemsetAsync(cellmap, 0xFF, (size_t)batch * D * hw * 4, st));
  if (n > 0)
    hipLaunchKernelGGL(k_cellmap_scatter, dim3(sv_grid_1d(n, DN_THREADS)), dim3(DN_THREADS), 0, st, reinterpret_cast<const int4*>(coords), n, batch, D, H, W,
                       cellmap);
  const size_t lds = (size_t)DH_POS * D * (C + 1) * sizeof(float);
  hipLaunchKernelGGL(k_dense_write_nhwc, dim3((unsigned)(batch * hw / DH_POS)), dim3(DN_THREADS), lds, st, features, cellmap, C, D, hw, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
extern "C" int sv_sparse_to_dense_nhwc_applies(int C, int D, int H, int W) {
  return (C > 0 && C % 4 == 0 && D >= 1 && D <= 8 && ((int64_t)H * W) % DH_POS == 0 && (size_t)DH_POS * D * (C + 1) * sizeof(float) <= 60 * 1024) ? 1 : 0;
}

// backward twin: grad_features[i][c] = grad[b][y][x][c D + d] for the voxel i at (b, d, y, x); a thread per (voxel, 4 channels): the lanes of a voxel walk
// its 4 C D-byte pixel row (every D-th float of it is theirs)
__global__ __launch_bounds__(DN_THREADS) void k_dense_gather_nhwc(const float* __restrict__ dense, const int4* __restrict__ coords, int64_t n, int B, int C,
                                                                  int D, int H, int W, float* __restrict__ out) {
  const int C4 = C / 4;
  const int64_t total = n * C4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / C4;
    const int c = (int)(idx - i * C4) * 4;
    const int4 p = coords[i];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.x >= 0 && p.x < B && p.y >= 0 && p.y < D && p.z >= 0 && p.z < H && p.w >= 0 && p.w < W) {
      const float* g = dense + (((int64_t)p.x * H + p.z) * W + p.w) * ((int64_t)C * D) + p.y;
      v = make_float4(g[(int64_t)c * D], g[(int64_t)(c + 1) * D], g[(int64_t)(c + 2) * D], g[(int64_t)(c + 3) * D]);
    }
    *reinterpret_cast<float4*>(out + i * C + c) = v;
  }
}
extern "C" int sv_dense_to_sparse_nhwc(const float* dense, const int32_t* coords, int64_t n, int batch, int C, int D, int H, int W, float* out, void* stream) {
  SV_CHECK_ARG(n >= 0 && batch > 0 && C > 0 && C % 4 == 0 && D > 0 && H > 0 && W > 0, "dense_to_sparse_nhwc: bad arguments (C %% 4 == 0)");
  if (n == 0) return SV_OK;
  SV_CHECK_ARG(dense && coords && out && (uintptr_t)out % 16 == 0, "dense_to_sparse_nhwc: null or misaligned pointer");
  hipLaunchKernelGGL(k_dense_gather_nhwc, dim3(sv_grid_1d(n * (C / 4), DN_THREADS, 256 * 16)), dim3(DN_THREADS), 0, sv_stream(stream), dense,
                     reinterpret_cast<const int4*>(coords), n, batch, C, D, H, W, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}


// ------------------------------------------------------------------------------------------------
// sum(x^2) and y = scale * x in ONE pass over x (n floats, n % 4 == 0): the forward of a mean-square loss over a dense BEV tensor that leaves the
// loss's gradient (2 / n) x behind as it reads x -- one read + one write of the tensor instead of a reduction pass in the forward and a
// read + write pass in the backward (bench.py's stand-in for the BEV backbone: 577 MB at the KITTI geometry, 16 scenes).  Deterministic: a fixed
// grid of MS_WGS workgroups, each a fixed-order tree, partials combined in order by sv_mean_square's second launch.
// ------------------------------------------------------------------------------------------------
constexpr int MS_WGS = 2048, MS_THREADS = 256;
__global__ __launch_bounds__(MS_THREADS) void k_square_sum_scale(const float4* __restrict__ x, int64_t n4, float scale, float4* __restrict__ y, float* __restrict__ partial) {
  __shared__ float s_red[MS_THREADS];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * MS_THREADS + threadIdx.x; i < n4; i += (int64_t)MS_WGS * MS_THREADS) {
    const float4 v = x[i];
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    if (y) y[i] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
  }
  s_red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = MS_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_red[threadIdx.x] += s_red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0];
}
__global__ __launch_bounds__(MS_THREADS) void k_square_sum_finish(const float* __restrict__ partial, float mul, float* __restrict__ out) {
  __shared__ double s_red[MS_THREADS];
  double acc = 0.0;
  for (int i = threadIdx.x; i < MS_WGS; i += MS_THREADS) acc += (double)partial[i];
  s_red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = MS_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_red[threadIdx.x] += s_red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (float)(s_red[0] * (double)mul);
}
extern "C" size_t sv_mean_square_scratch_bytes(void) { return MS_WGS * sizeof(float); }
// *out = mul * sum(x^2); y (nullable) = scale * x.  mean(x^2) and its gradient: mul = 1 / n, scale = 2 / n.
extern "C" int sv_mean_square(const float* x, int64_t n, float mul, float scale, float* y, float* out, void* scratch, void* stream) {
  SV_CHECK_ARG(n > 0 && n % 4 == 0 && x && out && scratch, "sv_mean_square: n must be a positive multiple of 4, pointers non-null");
  SV_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0, "sv_mean_square: 16-byte alignment");
  hipStream_t st = sv_stream(stream);
  hipLaunchKernelGGL(k_square_sum_scale, dim3(MS_WGS), dim3(MS_THREADS), 0, st, reinterpret_cast<const float4*>(x), n / 4, scale, reinterpret_cast<float4*>(y),
                     reinterpret_cast<float*>(scratch));
  hipLaunchKernelGGL(k_square_sum_finish, dim3(1), dim3(MS_THREADS), 0, st, reinterpret_cast<const float*>(scratch), mul, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
// x *= *g unless *g == 1 (a device scalar: the upstream gradient of a loss is 1 in a plain backward, and then nothing is touched)
__global__ __launch_bounds__(MS_THREADS) void k_scale_unless_one(float4* __restrict__ x, int64_t n4, const float* __restrict__ g) {
  const float s = *g;
  if (s == 1.f) return;
  for (int64_t i = (int64_t)blockIdx.x * MS_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * MS_THREADS) {
    float4 v = x[i];
    v.x *= s, v.y *= s, v.z *= s, v.w *= s;
    x[i] = v;
  }
}
extern "C" int sv_scale_by_device_scalar(float* x, int64_t n, const float* g, void* stream) {
  SV_CHECK_ARG(n > 0 && n % 4 == 0 && x && g && (uintptr_t)x % 16 == 0, "sv_scale_by_device_scalar: bad arguments");
  hipLaunchKernelGGL(k_scale_unless_one, dim3(MS_WGS), dim3(MS_THREADS), 0, sv_stream(stream), reinterpret_cast<float4*>(x), n / 4, g);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
