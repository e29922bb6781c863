// Sparse-convolution rulebook build on the coordinate index (HBM + integer bound).
//
// The rulebook is kept "output-major": nbr[k][o] = input row feeding output row o through kernel offset k
// (or -1).  That is the same information as spconv's indice_pairs[k] = {(in, out)} lists, but regular, so
// the convolution kernels gather without atomics and produce every output row exactly once.
//
//  * submanifold conv (SubMConv3d): output set = input set in the caller's row order;
//    nbr[k][i] = row j with coord[j] = coord[i] + (k - K/2)*dilation.
//  * regular / strided conv (SparseConv3d): out = (in + pad - k*dilation) / stride where divisible and inside
//    [0, out_shape); the output set is the unique set of those coordinates in CANONICAL order = ascending
//    linear key ((b*Z + z)*Y + y)*X + x (spconv's own order is implementation-defined).
//
// spconv is an un-vendored third-party dependency of the reference (docker/Dockerfile:58); call sites replaced:
// detector3d/pcdet/models/backbones_3d/spconv_backbone.py:77-117 (layer definitions), :141-157 (forward).
#include "common.h"

struct ConvGeom {
  int batch;
  int in_shape[3];   // Z, Y, X
  int out_shape[3];
  int ksize[3];
  int stride[3];
  int pad[3];
  int dil[3];
  int K;
};

constexpr int RB_THREADS = 256;

__device__ __forceinline__ int64_t lin_key(int b, int z, int y, int x, const int* shape) {
  return (((int64_t)b * shape[0] + z) * shape[1] + y) * shape[2] + x;
}

__device__ __forceinline__ bool coord_ok(const int4 c, int batch, const int* shape) {
  return c.x >= 0 && c.x < batch && c.y >= 0 && c.y < shape[0] && c.z >= 0 && c.z < shape[1] && c.w >= 0 && c.w < shape[2];
}

// --------------------------------------------------------------------------- index finalisation
// For every occupied chunk: words[w].y = chunk_base + popcount of the chunk's earlier words.
// One wave inspects 64 chunk counters at a time and walks the occupied ones with 32 lanes per chunk.
__global__ __launch_bounds__(RB_THREADS) void k_index_prefix(SvIndexView ix, int64_t nchunks) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t c0 = wave * 64; c0 < nchunks; c0 += nwaves * 64) {
    const int64_t c = c0 + lane;
    const int cnt = c < nchunks ? ix.chunk_cnt[c] : 0;
    unsigned long long occ = __ballot(cnt > 0);
    while (occ) {
      // two occupied chunks per step: lanes 0-31 take the first, lanes 32-63 the second
      const int first = __ffsll((long long)occ) - 1;
      occ &= occ - 1;
      int second = -1;
      if (occ) { second = __ffsll((long long)occ) - 1; occ &= occ - 1; }
      const int mine = lane < 32 ? first : second;
      if (mine >= 0) {
        const int64_t cc = c0 + mine;
        const int64_t w = cc * SV_CHUNK_WORDS + (lane & 31);
        const uint32_t bits = ix.words[w].x;
        int p = __popc(bits);
        int incl = p;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
          const int t = __shfl_up(incl, d, 32);
          if ((lane & 31) >= d) incl += t;
        }
        if (bits) ix.words[w].y = (uint32_t)(ix.chunk_base[cc] + incl - p);
      }
    }
  }
}

static int launch_index_prefix(const SvIndexView& ix, hipStream_t st) {
  const int64_t nchunks = sv_index_nchunks(ix.ncells);
  const int grid = sv_grid_1d((nchunks + 63) / 64 * 64, RB_THREADS);
  hipLaunchKernelGGL(k_index_prefix, dim3(grid), dim3(RB_THREADS), 0, st, ix, nchunks);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// --------------------------------------------------------------------------- submanifold
__global__ __launch_bounds__(RB_THREADS) void k_subm_mark(const int4* __restrict__ coords, int64_t n, ConvGeom g, SvIndexView ix) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = coords[i];
    if (coord_ok(c, g.batch, g.in_shape)) sv_index_mark(ix, lin_key(c.x, c.y, c.z, c.w, g.in_shape));
  }
}

__global__ __launch_bounds__(RB_THREADS) void k_subm_perm(const int4* __restrict__ coords, int64_t n, ConvGeom g, SvIndexView ix,
                                                          int32_t* __restrict__ perm) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = coords[i];
    if (coord_ok(c, g.batch, g.in_shape)) {
      const int32_t r = sv_index_lookup(ix, lin_key(c.x, c.y, c.z, c.w, g.in_shape));
      perm[r] = (int32_t)i;
    }
  }
}

// one thread per (row, kernel offset); offsets vary fastest across a row's lanes is NOT wanted (stores to
// nbr[k][i] must be coalesced over i), so the grid is laid out k-major: idx = k*n + i.
__global__ __launch_bounds__(RB_THREADS) void k_subm_query(const int4* __restrict__ coords, int64_t n, ConvGeom g, SvIndexView ix,
                                                           const int32_t* __restrict__ perm, int32_t* __restrict__ nbr) {
  const int64_t total = n * g.K;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx / n);
    const int64_t i = idx - (int64_t)k * n;
    const int kx = k % g.ksize[2], ky = (k / g.ksize[2]) % g.ksize[1], kz = k / (g.ksize[2] * g.ksize[1]);
    const int4 c = coords[i];
    int32_t r = -1;
    if (coord_ok(c, g.batch, g.in_shape)) {
      const int z = c.y + (kz - g.ksize[0] / 2) * g.dil[0];
      const int y = c.z + (ky - g.ksize[1] / 2) * g.dil[1];
      const int x = c.w + (kx - g.ksize[2] / 2) * g.dil[2];
      if (z >= 0 && z < g.in_shape[0] && y >= 0 && y < g.in_shape[1] && x >= 0 && x < g.in_shape[2]) {
        const int32_t rank = sv_index_lookup(ix, lin_key(c.x, z, y, x, g.in_shape));
        if (rank >= 0) r = perm[rank];
      }
    }
    nbr[idx] = r;
  }
}

__global__ __launch_bounds__(RB_THREADS) void k_subm_clear(const int4* __restrict__ coords, int64_t n, ConvGeom g, SvIndexView ix) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = coords[i];
    if (coord_ok(c, g.batch, g.in_shape)) {
      const int64_t key = lin_key(c.x, c.y, c.z, c.w, g.in_shape);
      ix.words[key >> 5] = make_uint2(0u, 0u);
      ix.chunk_cnt[key >> SV_CHUNK_SHIFT] = 0;
    }
  }
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

extern "C" size_t sv_rulebook_scratch_bytes(int64_t n_in, int64_t ncells) {
  return align256((size_t)(n_in > 0 ? n_in : 1) * 4) + align256(sv_index_scan_tmp_bytes(ncells)) + 256;
}

static int fill_geom(ConvGeom& g, int batch, const int32_t* in_shape, const int32_t* ksize, const int32_t* stride,
                     const int32_t* pad, const int32_t* dil, bool subm) {
  g.batch = batch;
  g.K = 1;
  for (int d = 0; d < 3; ++d) {
    g.in_shape[d] = in_shape[d];
    g.ksize[d] = ksize[d];
    g.stride[d] = subm ? 1 : stride[d];
    g.pad[d] = subm ? (ksize[d] / 2) * (dil ? dil[d] : 1) : pad[d];
    g.dil[d] = dil ? dil[d] : 1;
    SV_CHECK_ARG(g.in_shape[d] > 0 && g.ksize[d] > 0 && g.stride[d] > 0 && g.pad[d] >= 0 && g.dil[d] > 0, "rulebook: bad conv geometry");
    // floor((D + 2p - dil*(K-1) - 1)/s) + 1
    g.out_shape[d] = subm ? g.in_shape[d] : (g.in_shape[d] + 2 * g.pad[d] - g.dil[d] * (g.ksize[d] - 1) - 1) / g.stride[d] + 1;
    SV_CHECK_ARG(g.out_shape[d] > 0, "rulebook: empty output shape");
    g.K *= g.ksize[d];
  }
  return SV_OK;
}

extern "C" int sv_rulebook_subm(const int32_t* coords, int64_t n, int batch, const int32_t* shape_host,
                                const int32_t* ksize_host, const int32_t* dilation_host, void* index_ws, void* scratch,
                                int32_t* nbr, void* stream) {
  SV_CHECK_ARG(n >= 0 && batch > 0 && shape_host && ksize_host, "rulebook_subm: bad arguments");
  if (n == 0) return SV_OK;
  SV_CHECK_ARG(coords && index_ws && scratch && nbr, "rulebook_subm: null pointer");
  ConvGeom g;
  int rc = fill_geom(g, batch, shape_host, ksize_host, nullptr, nullptr, dilation_host, true);
  if (rc) return rc;
  SV_CHECK_ARG((g.ksize[0] & 1) && (g.ksize[1] & 1) && (g.ksize[2] & 1), "rulebook_subm: kernel sizes must be odd");
  hipStream_t st = sv_stream(stream);
  const int64_t ncells = (int64_t)batch * g.in_shape[0] * g.in_shape[1] * g.in_shape[2];
  SvIndexView ix = sv_index_view(index_ws, ncells);
  char* s = reinterpret_cast<char*>(scratch);
  int32_t* perm = reinterpret_cast<int32_t*>(s);
  s += align256((size_t)n * 4);
  void* scan_tmp = s;
  s += align256(sv_index_scan_tmp_bytes(ncells));
  int32_t* total = reinterpret_cast<int32_t*>(s);
  const int4* c4 = reinterpret_cast<const int4*>(coords);
  const int grid = sv_grid_1d(n, RB_THREADS);
  hipLaunchKernelGGL(k_subm_mark, dim3(grid), dim3(RB_THREADS), 0, st, c4, n, g, ix);
  rc = sv_index_scan_launch(ix, total, scan_tmp, st);
  if (rc) return rc;
  rc = launch_index_prefix(ix, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_subm_perm, dim3(grid), dim3(RB_THREADS), 0, st, c4, n, g, ix, perm);
  hipLaunchKernelGGL(k_subm_query, dim3(sv_grid_1d(n * g.K, RB_THREADS, 256 * 16)), dim3(RB_THREADS), 0, st, c4, n, g, ix, perm, nbr);
  hipLaunchKernelGGL(k_subm_clear, dim3(grid), dim3(RB_THREADS), 0, st, c4, n, g, ix);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// --------------------------------------------------------------------------- submanifold through a dense cell -> row map
// 288 GB of HBM: a persistent int32 per cell (5.9 GB for 16 KITTI scenes at 5 cm) turns the submanifold rulebook into three
// passes -- store row+1 at the voxel's cell, read the 27 neighbour cells, store 0 again -- without the rank dictionary's
// atomics, scan, prefix and permutation passes (8 launches -> 3).  The map holds 0 in every cell between calls.
// The map is tiled 4 x 4 x 8 cells (512 bytes): the 27 neighbours of a voxel fall into one or two tiles and consecutive rows stay in
// the same tiles whether they arrive z-fastest (the voxeliser's order) or x-fastest (a strided conv's output order).  With a plain
// x-fastest layout the query of the 213 k-voxel input layer pulled 262 MB from HBM (PMC FETCH_SIZE), 12 lines per row.
struct CellTiling {
  int tz, ty, tx;      // tiles per axis
};
__device__ __forceinline__ int64_t tiled_cell(int b, int z, int y, int x, const CellTiling& t) {
  const int64_t tile = (((int64_t)b * t.tz + (z >> 2)) * t.ty + (y >> 2)) * t.tx + (x >> 3);
  return tile * 128 + (((z & 3) * 4 + (y & 3)) * 8 + (x & 7));
}
static CellTiling cell_tiling(const int* shape) { return CellTiling{(shape[0] + 3) / 4, (shape[1] + 3) / 4, (shape[2] + 7) / 8}; }

__global__ __launch_bounds__(RB_THREADS) void k_cellmap_set(const int4* __restrict__ coords, int64_t n, ConvGeom g, CellTiling t,
                                                            int32_t* __restrict__ map, int clear) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = coords[i];
    if (coord_ok(c, g.batch, g.in_shape)) map[tiled_cell(c.x, c.y, c.z, c.w, t)] = clear ? 0 : (int32_t)i + 1;
  }
}

__global__ __launch_bounds__(RB_THREADS) void k_subm_query_map(const int4* __restrict__ coords, int64_t n, ConvGeom g, CellTiling t,
                                                               const int32_t* __restrict__ map, int32_t* __restrict__ nbr) {
  const int64_t total = n * g.K;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx / n);                 // k-major: the stores to nbr[k][i] are coalesced over i
    const int64_t i = idx - (int64_t)k * n;
    const int kx = k % g.ksize[2], ky = (k / g.ksize[2]) % g.ksize[1], kz = k / (g.ksize[2] * g.ksize[1]);
    const int4 c = coords[i];
    int32_t r = -1;
    if (coord_ok(c, g.batch, g.in_shape)) {
      const int z = c.y + (kz - g.ksize[0] / 2) * g.dil[0];
      const int y = c.z + (ky - g.ksize[1] / 2) * g.dil[1];
      const int x = c.w + (kx - g.ksize[2] / 2) * g.dil[2];
      if (z >= 0 && z < g.in_shape[0] && y >= 0 && y < g.in_shape[1] && x >= 0 && x < g.in_shape[2])
        r = map[tiled_cell(c.x, z, y, x, t)] - 1;
    }
    nbr[idx] = r;
  }
}

// One thread per ROW (K <= 27): the 27 neighbour cells are read as independent loads, the k-major table is written with stores that are
// coalesced across the rows of a wave, and the same values go out once more ROW-MAJOR (128 bytes per row: [0..26] source rows, the rest -1) with
// the row's neighbour mask -- what the convolution's plan (sparse_conv.hip) consumes: a 16-row tile reads 16 lines instead of 27 x 16 words.
constexpr int RB_ROW = 32;     // int32 per row of a row-major table (== PL_ROW in sparse_conv.hip)
constexpr int RB_KMAX = 27;
typedef int rb_i32x4 __attribute__((ext_vector_type(4)));

template <bool FULL3>   // FULL3: 3x3x3 kernel, offsets decomposed at compile time
__global__ __launch_bounds__(RB_THREADS) void k_subm_query_rows(const int4* __restrict__ coords, int64_t n, ConvGeom g, CellTiling t,
                                                                const int32_t* __restrict__ map, int32_t* __restrict__ nbr,
                                                                int32_t* __restrict__ tab, int32_t* __restrict__ masks) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = coords[i];
  const bool ok = coord_ok(c, g.batch, g.in_shape);
  const int hz = g.ksize[0] / 2, hy = g.ksize[1] / 2, hx = g.ksize[2] / 2, kyx = g.ksize[1] * g.ksize[2];
  int32_t e[RB_ROW];
#pragma unroll
  for (int k = 0; k < RB_ROW; ++k) e[k] = -1;
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < RB_KMAX; ++k) {
    if (k < g.K) {
      const int kz = FULL3 ? k / 9 : k / kyx, ky = FULL3 ? (k / 3) % 3 : (k - kz * kyx) / g.ksize[2], kx = FULL3 ? k % 3 : k - kz * kyx - ky * g.ksize[2];
      const int z = c.y + (kz - hz) * g.dil[0], y = c.z + (ky - hy) * g.dil[1], x = c.w + (kx - hx) * g.dil[2];
      int32_t r = -1;
      if (ok && z >= 0 && z < g.in_shape[0] && y >= 0 && y < g.in_shape[1] && x >= 0 && x < g.in_shape[2]) r = map[tiled_cell(c.x, z, y, x, t)] - 1;
      e[k] = r;
    }
  }
#pragma unroll
  for (int k = 0; k < RB_KMAX; ++k) {
    if (k < g.K) {
      nbr[(int64_t)k * n + i] = e[k];
      m |= e[k] >= 0 ? (1u << k) : 0u;
    }
  }
  if (masks) masks[i] = (int32_t)m;
  if (tab) {
    rb_i32x4* dst = reinterpret_cast<rb_i32x4*>(tab + i * RB_ROW);
#pragma unroll
    for (int q = 0; q < RB_ROW / 4; ++q) dst[q] = (rb_i32x4){e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]};
  }
}

extern "C" size_t sv_cellmap_persistent_bytes(int batch, const int32_t* spatial_shape) {
  if (batch <= 0 || !spatial_shape) return 0;
  const CellTiling t = cell_tiling(spatial_shape);
  return (size_t)batch * t.tz * t.ty * t.tx * 128 * sizeof(int32_t);
}

extern "C" int sv_rulebook_subm_cellmap(const int32_t* coords, int64_t n, int batch, const int32_t* shape_host, const int32_t* ksize_host,
                                        const int32_t* dilation_host, void* cellmap, int32_t* nbr, int32_t* table_rows, int32_t* masks, void* stream) {
  SV_CHECK_ARG(n >= 0 && batch > 0 && shape_host && ksize_host, "rulebook_subm_cellmap: bad arguments");
  SV_CHECK_ARG(n < 0x7fffffff, "rulebook_subm_cellmap: row + 1 must fit an int32");
  if (n == 0) return SV_OK;
  SV_CHECK_ARG(coords && cellmap && nbr, "rulebook_subm_cellmap: null pointer");
  ConvGeom g;
  int rc = fill_geom(g, batch, shape_host, ksize_host, nullptr, nullptr, dilation_host, true);
  if (rc) return rc;
  SV_CHECK_ARG((g.ksize[0] & 1) && (g.ksize[1] & 1) && (g.ksize[2] & 1), "rulebook_subm_cellmap: kernel sizes must be odd");
  SV_CHECK_ARG(g.K <= RB_KMAX || (!table_rows && !masks), "rulebook_subm_cellmap: row-major table / masks need K <= %d", RB_KMAX);
  hipStream_t st = sv_stream(stream);
  const int4* c4 = reinterpret_cast<const int4*>(coords);
  int32_t* map = reinterpret_cast<int32_t*>(cellmap);
  const int grid = sv_grid_1d(n, RB_THREADS);
  const CellTiling t = cell_tiling(g.in_shape);
  hipLaunchKernelGGL(k_cellmap_set, dim3(grid), dim3(RB_THREADS), 0, st, c4, n, g, t, map, 0);
  if (g.ksize[0] == 3 && g.ksize[1] == 3 && g.ksize[2] == 3)
    hipLaunchKernelGGL(k_subm_query_rows<true>, dim3(sv_div_up(n, RB_THREADS)), dim3(RB_THREADS), 0, st, c4, n, g, t, map, nbr, table_rows, masks);
  else if (g.K <= RB_KMAX)
    hipLaunchKernelGGL(k_subm_query_rows<false>, dim3(sv_div_up(n, RB_THREADS)), dim3(RB_THREADS), 0, st, c4, n, g, t, map, nbr, table_rows, masks);
  else
    hipLaunchKernelGGL(k_subm_query_map, dim3(sv_grid_1d(n * g.K, RB_THREADS, 256 * 16)), dim3(RB_THREADS), 0, st, c4, n, g, t, map, nbr);
  hipLaunchKernelGGL(k_cellmap_set, dim3(grid), dim3(RB_THREADS), 0, st, c4, n, g, t, map, 1);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// --------------------------------------------------------------------------- regular / strided
// candidate output coordinate of input c through offset (kz,ky,kx); false when not on the output lattice
__device__ __forceinline__ bool out_coord(const int4 c, int kz, int ky, int kx, const ConvGeom& g, int& oz, int& oy, int& ox) {
  const int tz = c.y + g.pad[0] - kz * g.dil[0];
  const int ty = c.z + g.pad[1] - ky * g.dil[1];
  const int tx = c.w + g.pad[2] - kx * g.dil[2];
  if (tz < 0 || ty < 0 || tx < 0) return false;
  if (tz % g.stride[0] || ty % g.stride[1] || tx % g.stride[2]) return false;
  oz = tz / g.stride[0]; oy = ty / g.stride[1]; ox = tx / g.stride[2];
  return oz < g.out_shape[0] && oy < g.out_shape[1] && ox < g.out_shape[2];
}

template <int PHASE>  // 0: mark output cells; 1: write nbr_in + out_coords; 2: clear
__global__ __launch_bounds__(RB_THREADS) void k_sparse_phase(const int4* __restrict__ coords, int64_t n, ConvGeom g, SvIndexView ix,
                                                             int32_t* __restrict__ nbr_in, int4* __restrict__ out_coords,
                                                             int64_t capacity) {
  const int64_t total = n * g.K;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx / n);
    const int64_t i = idx - (int64_t)k * n;
    const int kx = k % g.ksize[2], ky = (k / g.ksize[2]) % g.ksize[1], kz = k / (g.ksize[2] * g.ksize[1]);
    const int4 c = coords[i];
    int oz, oy, ox;
    int32_t r = -1;
    if (coord_ok(c, g.batch, g.in_shape) && out_coord(c, kz, ky, kx, g, oz, oy, ox)) {
      const int64_t key = lin_key(c.x, oz, oy, ox, g.out_shape);
      if (PHASE == 0) {
        sv_index_mark(ix, key);
      } else if (PHASE == 1) {
        r = sv_index_lookup(ix, key);
        if (r >= capacity) r = -1;
        if (r >= 0) out_coords[r] = make_int4(c.x, oz, oy, ox);  // identical value from every contributor
      } else {
        ix.words[key >> 5] = make_uint2(0u, 0u);
        ix.chunk_cnt[key >> SV_CHUNK_SHIFT] = 0;
      }
    }
    if (PHASE == 1) nbr_in[idx] = r;
  }
}

// ---- one thread per INPUT row (K <= 27).  Marking needs no answer from memory: non-returning atomicOr on the occupancy words; the number
// of occupied cells per chunk and the prefix inside a chunk then come from ONE streaming pass over the bitmap (k_index_count: 8 bytes per
// 32 cells; 47 MB for 16 KITTI scenes at stride 2) instead of a returning atomic per candidate plus a separate prefix pass over the occupied
// chunks.  Measured before (bench scenes, four strided layers per step): mark 62 us (returning atomics), prefix 26 us.
// Per-axis candidates of one input coordinate: for kernel index ka along the axis, the output coordinate it feeds (or -1).  3 divisions per
// axis (none for stride 1 or 2) instead of 3 per kernel offset: the 27 offsets of a 3x3x3 kernel are the products of these.
struct AxisCand {
  int o[3][3];       // [axis][ka] output coordinate or -1  (kernel sizes up to 3 per axis)
};
__device__ __forceinline__ AxisCand axis_candidates(const int4 c, const ConvGeom& g) {
  AxisCand a;
  const int cc[3] = {c.y, c.z, c.w};
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const int s = g.stride[ax];
#pragma unroll
    for (int ka = 0; ka < 3; ++ka) {
      int o = -1;
      if (ka < g.ksize[ax]) {
        const int t = cc[ax] + g.pad[ax] - ka * g.dil[ax];
        if (t >= 0) {
          if (s == 1) o = t;
          else if (s == 2) o = (t & 1) ? -1 : (t >> 1);
          else o = (t % s) ? -1 : t / s;
          if (o >= g.out_shape[ax]) o = -1;
        }
      }
      a.o[ax][ka] = o;
    }
  }
  return a;
}
__device__ __forceinline__ bool small_kernel(const ConvGeom& g) { return g.ksize[0] <= 3 && g.ksize[1] <= 3 && g.ksize[2] <= 3; }

// every output cell input coordinate c reaches: set its bit (no-return atomic) or, clear != 0, return its word and chunk count to zero
template <bool FULL3>   // FULL3: 3x3x3 kernel, offsets decomposed at compile time
__device__ __forceinline__ void sparse_mark_one(const int4 c, const ConvGeom& g, const SvIndexView& ix, int clear) {
  const int kyx = g.ksize[1] * g.ksize[2];
  const bool small = FULL3 || small_kernel(g);            // wave-uniform
  const AxisCand ac = small ? axis_candidates(c, g) : AxisCand{};
#pragma unroll
  for (int k = 0; k < RB_KMAX; ++k) {
    if (k < g.K) {
      const int kz = FULL3 ? k / 9 : k / kyx, ky = FULL3 ? (k / 3) % 3 : (k - kz * kyx) / g.ksize[2], kx = FULL3 ? k % 3 : k - kz * kyx - ky * g.ksize[2];
      int oz, oy, ox;
      bool hit;
      if (small) {
        oz = ac.o[0][kz % 3], oy = ac.o[1][ky % 3], ox = ac.o[2][kx % 3];
        hit = (oz | oy | ox) >= 0;
      } else {
        hit = out_coord(c, kz, ky, kx, g, oz, oy, ox);
      }
      if (hit) {
        const int64_t key = lin_key(c.x, oz, oy, ox, g.out_shape);
        if (clear) {
          ix.words[key >> 5] = make_uint2(0u, 0u);
          ix.chunk_cnt[key >> SV_CHUNK_SHIFT] = 0;
        } else {
          __hip_atomic_fetch_or(&ix.words[key >> 5].x, 1u << (key & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // result unused: no-return atomic
        }
      }
    }
  }
}

template <bool FULL3>
__global__ __launch_bounds__(RB_THREADS) void k_sparse_mark_rows(const int4* __restrict__ coords, int64_t n, ConvGeom g, SvIndexView ix, int clear) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = coords[i];
  if (!coord_ok(c, g.batch, g.in_shape)) return;
  sparse_mark_one<FULL3>(c, g, ix, clear);
}

// chunk_cnt[c] = occupied cells of chunk c; words[w].y = occupied cells of the chunk before word w (only written where the word is
// non-empty, so untouched words stay all-zero).  32 lanes per chunk.
__global__ __launch_bounds__(RB_THREADS) void k_index_count(SvIndexView ix, int64_t nchunks) {
  const int lane = threadIdx.x & 31;
  const int64_t sub = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5, nsub = ((int64_t)gridDim.x * blockDim.x) >> 5;
  for (int64_t c = sub; c < nchunks; c += nsub) {
    const int64_t w = c * SV_CHUNK_WORDS + lane;
    const uint32_t bits = ix.words[w].x;
    const int p = __popc(bits);
    int incl = p;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int t = __shfl_up(incl, d, 32);
      if (lane >= d) incl += t;
    }
    if (bits) ix.words[w].y = (uint32_t)(incl - p);
    if (lane == 31) ix.chunk_cnt[c] = incl;
  }
}

// rank of key or -1: chunk base + prefix inside the chunk + bits below (the in-chunk prefixes of k_index_count)
__device__ __forceinline__ int32_t index_lookup_chunked(const SvIndexView& ix, int64_t key) {
  const uint2 wd = ix.words[key >> 5];
  const uint32_t bit = 1u << (key & 31);
  if (!(wd.x & bit)) return -1;
  return ix.chunk_base[key >> SV_CHUNK_SHIFT] + (int32_t)wd.y + __popc(wd.x & (bit - 1u));
}

template <bool FULL3>
__global__ __launch_bounds__(RB_THREADS) void k_sparse_lookup_rows(const int4* __restrict__ coords, int64_t n, ConvGeom g, SvIndexView ix,
                                                                   int32_t* __restrict__ nbr_in, int4* __restrict__ out_coords, int64_t capacity,
                                                                   int32_t* __restrict__ tab_in, int32_t* __restrict__ masks_in) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = coords[i];
  const bool ok = coord_ok(c, g.batch, g.in_shape);
  const int kyx = g.ksize[1] * g.ksize[2];
  const bool small = FULL3 || small_kernel(g);            // wave-uniform
  const AxisCand ac = small ? axis_candidates(c, g) : AxisCand{};
  int32_t e[RB_ROW];
#pragma unroll
  for (int k = 0; k < RB_ROW; ++k) e[k] = -1;
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < RB_KMAX; ++k) {
    if (k < g.K) {
      const int kz = FULL3 ? k / 9 : k / kyx, ky = FULL3 ? (k / 3) % 3 : (k - kz * kyx) / g.ksize[2], kx = FULL3 ? k % 3 : k - kz * kyx - ky * g.ksize[2];
      int oz, oy, ox;
      bool hit;
      if (small) {
        oz = ac.o[0][kz % 3], oy = ac.o[1][ky % 3], ox = ac.o[2][kx % 3];
        hit = (oz | oy | ox) >= 0;
      } else {
        hit = out_coord(c, kz, ky, kx, g, oz, oy, ox);
      }
      if (ok && hit) {
        int32_t r = index_lookup_chunked(ix, lin_key(c.x, oz, oy, ox, g.out_shape));
        if (r >= capacity) r = -1;
        if (r >= 0) {
          out_coords[r] = make_int4(c.x, oz, oy, ox);      // identical value from every contributor
          m |= 1u << k;
        }
        e[k] = r;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < RB_KMAX; ++k)
    if (k < g.K) nbr_in[(int64_t)k * n + i] = e[k];
  if (masks_in) masks_in[i] = (int32_t)m;
  if (tab_in) {
    rb_i32x4* dst = reinterpret_cast<rb_i32x4*>(tab_in + i * RB_ROW);
#pragma unroll
    for (int q = 0; q < RB_ROW / 4; ++q) dst[q] = (rb_i32x4){e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]};
  }
}

extern "C" int sv_conv_out_shape(const int32_t* in_shape_host, const int32_t* ksize_host, const int32_t* stride_host,
                                 const int32_t* padding_host, const int32_t* dilation_host, int32_t* out_shape_host) {
  ConvGeom g;
  int rc = fill_geom(g, 1, in_shape_host, ksize_host, stride_host, padding_host, dilation_host, false);
  if (rc) return rc;
  for (int d = 0; d < 3; ++d) out_shape_host[d] = g.out_shape[d];
  return SV_OK;
}

extern "C" int sv_rulebook_sparse(const int32_t* coords, int64_t n_in, int batch, const int32_t* in_shape_host,
                                  const int32_t* ksize_host, const int32_t* stride_host, const int32_t* padding_host,
                                  const int32_t* dilation_host, void* index_ws, void* scratch, int32_t* out_coords,
                                  int32_t* nbr_in, int32_t* in_block, int64_t capacity, int32_t* num_out, void* stream) {
  SV_CHECK_ARG(n_in >= 0 && batch > 0 && capacity >= 0 && num_out, "rulebook_sparse: bad arguments");
  hipStream_t st = sv_stream(stream);
  if (n_in == 0) {
    SV_HIP(hipMemsetAsync(num_out, 0, 4, st));
    return SV_OK;
  }
  SV_CHECK_ARG(coords && index_ws && scratch && out_coords && nbr_in, "rulebook_sparse: null pointer");
  ConvGeom g;
  int rc = fill_geom(g, batch, in_shape_host, ksize_host, stride_host, padding_host, dilation_host, false);
  if (rc) return rc;
  SV_CHECK_ARG(g.K <= RB_KMAX || !in_block, "rulebook_sparse: the row-major input table needs K <= %d", RB_KMAX);
  const int64_t ncells = (int64_t)batch * g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
  SvIndexView ix = sv_index_view(index_ws, ncells);
  char* s = reinterpret_cast<char*>(scratch) + align256((size_t)n_in * 4);
  void* scan_tmp = s;
  const int4* c4 = reinterpret_cast<const int4*>(coords);
  int4* oc4 = reinterpret_cast<int4*>(out_coords);
  if (g.K <= RB_KMAX) {
    // rows: mark (no-return atomics) -> count the whole bitmap -> scan the chunk counts -> look up, write both input-major tables -> clear
    const int rows_grid = sv_div_up(n_in, RB_THREADS);
    const int64_t nchunks = sv_index_nchunks(ncells);
    const bool full3 = g.ksize[0] == 3 && g.ksize[1] == 3 && g.ksize[2] == 3;
    if (full3) hipLaunchKernelGGL(k_sparse_mark_rows<true>, dim3(rows_grid), dim3(RB_THREADS), 0, st, c4, n_in, g, ix, 0);
    else hipLaunchKernelGGL(k_sparse_mark_rows<false>, dim3(rows_grid), dim3(RB_THREADS), 0, st, c4, n_in, g, ix, 0);
    hipLaunchKernelGGL(k_index_count, dim3(sv_grid_1d(nchunks * 32, RB_THREADS, 256 * 16)), dim3(RB_THREADS), 0, st, ix, nchunks);
    rc = sv_index_scan_launch(ix, num_out, scan_tmp, st);
    if (rc) return rc;
    int32_t* masks_in = in_block ? in_block + (size_t)RB_ROW * n_in : nullptr;
    if (full3) hipLaunchKernelGGL(k_sparse_lookup_rows<true>, dim3(rows_grid), dim3(RB_THREADS), 0, st, c4, n_in, g, ix, nbr_in, oc4, capacity, in_block, masks_in);
    else hipLaunchKernelGGL(k_sparse_lookup_rows<false>, dim3(rows_grid), dim3(RB_THREADS), 0, st, c4, n_in, g, ix, nbr_in, oc4, capacity, in_block, masks_in);
    if (full3) hipLaunchKernelGGL(k_sparse_mark_rows<true>, dim3(rows_grid), dim3(RB_THREADS), 0, st, c4, n_in, g, ix, 1);
    else hipLaunchKernelGGL(k_sparse_mark_rows<false>, dim3(rows_grid), dim3(RB_THREADS), 0, st, c4, n_in, g, ix, 1);
    SV_LAUNCH_CHECK();
    return SV_OK;
  }
  const int grid = sv_grid_1d(n_in * g.K, RB_THREADS, 256 * 16);
  hipLaunchKernelGGL(k_sparse_phase<0>, dim3(grid), dim3(RB_THREADS), 0, st, c4, n_in, g, ix, nbr_in, oc4, capacity);
  rc = sv_index_scan_launch(ix, num_out, scan_tmp, st);
  if (rc) return rc;
  rc = launch_index_prefix(ix, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sparse_phase<1>, dim3(grid), dim3(RB_THREADS), 0, st, c4, n_in, g, ix, nbr_in, oc4, capacity);
  hipLaunchKernelGGL(k_sparse_phase<2>, dim3(grid), dim3(RB_THREADS), 0, st, c4, n_in, g, ix, nbr_in, oc4, capacity);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// nbr_in (K, n_in) [input-major: output row per (k, input)]  ->  nbr_out (K, n_out) [output-major]
__global__ __launch_bounds__(RB_THREADS) void k_invert(const int32_t* __restrict__ nbr_in, int64_t n_in, int K, int32_t* __restrict__ nbr_out,
                                                       int64_t n_out) {
  const int64_t total = n_in * K;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int32_t o = nbr_in[idx];
    if (o >= 0 && o < n_out) {
      const int k = (int)(idx / n_in);
      nbr_out[(int64_t)k * n_out + o] = (int32_t)(idx - (int64_t)k * n_in);  // unique writer: (k, o) fixes the input coord
    }
  }
}

// The same inversion with one thread per INPUT row (K <= 27) from the row-major input table (one 128-byte line per thread instead of 27
// strided words); additionally writes the row-major twin and the neighbour masks of the OUTPUT side -- what the convolution plans
// consume -- by scattered stores / atomicOr (tab_out pre-filled with -1, masks_out with 0).
__global__ __launch_bounds__(RB_THREADS) void k_invert_rows(const int32_t* __restrict__ tab_in, int64_t n_in, int K, int32_t* __restrict__ nbr_out,
                                                            int64_t n_out, int32_t* __restrict__ tab_out, int32_t* __restrict__ masks_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_in) return;
  const rb_i32x4* src = reinterpret_cast<const rb_i32x4*>(tab_in + i * RB_ROW);
  rb_i32x4 v[(RB_KMAX + 3) / 4];
#pragma unroll
  for (int q = 0; q < (RB_KMAX + 3) / 4; ++q) v[q] = src[q];
#pragma unroll
  for (int k = 0; k < RB_KMAX; ++k) {
    const int32_t o = v[k >> 2][k & 3];
    if (k < K && o >= 0 && o < n_out) {
      nbr_out[(int64_t)k * n_out + o] = (int32_t)i;          // unique writer: (k, o) fixes the input coord
      tab_out[(int64_t)o * RB_ROW + k] = (int32_t)i;
      __hip_atomic_fetch_or(reinterpret_cast<unsigned*>(masks_out) + o, 1u << k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

extern "C" int sv_rulebook_invert(const int32_t* nbr_in, int64_t n_in, int K, int32_t* nbr_out, int64_t n_out, void* stream) {
  SV_CHECK_ARG(n_in >= 0 && n_out >= 0 && K > 0, "rulebook_invert: bad arguments");
  if (n_out == 0) return SV_OK;
  SV_CHECK_ARG(nbr_out, "rulebook_invert: null pointer");
  hipStream_t st = sv_stream(stream);
  SV_HIP(hipMemsetAsync(nbr_out, 0xFF, (size_t)K * n_out * 4, st));
  if (n_in == 0) return SV_OK;
  SV_CHECK_ARG(nbr_in, "rulebook_invert: null pointer");
  hipLaunchKernelGGL(k_invert, dim3(sv_grid_1d(n_in * K, RB_THREADS, 256 * 16)), dim3(RB_THREADS), 0, st, nbr_in, n_in, K, nbr_out, n_out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// out_block: ONE allocation of (32 * n_out + K * n_out + n_out) int32 = [tab_out (n_out, 32) | nbr_out (K, n_out) | masks_out (n_out)]
// (filled here with two memsets); table_rows_in (n_in, 32): the row-major input table of sv_rulebook_sparse (in_block).
extern "C" int sv_rulebook_invert_rows(const int32_t* table_rows_in, int64_t n_in, int K, int32_t* out_block, int64_t n_out, void* stream) {
  SV_CHECK_ARG(n_in >= 0 && n_out >= 0 && K > 0 && K <= RB_KMAX, "rulebook_invert_rows: 1 <= K <= %d", RB_KMAX);
  hipStream_t st = sv_stream(stream);
  if (n_out == 0) return SV_OK;
  SV_CHECK_ARG(out_block, "rulebook_invert_rows: null pointer");
  SV_HIP(hipMemsetAsync(out_block, 0xFF, (size_t)(K + RB_ROW) * n_out * 4, st));
  SV_HIP(hipMemsetAsync(out_block + (size_t)(K + RB_ROW) * n_out, 0, (size_t)n_out * 4, st));
  if (n_in == 0) return SV_OK;
  SV_CHECK_ARG(table_rows_in, "rulebook_invert_rows: null pointer");
  int32_t* tab_out = out_block;                                   // first: 16-byte aligned rows
  int32_t* nbr_out = out_block + (size_t)RB_ROW * n_out;
  int32_t* masks_out = nbr_out + (size_t)K * n_out;
  hipLaunchKernelGGL(k_invert_rows, dim3(sv_div_up(n_in, RB_THREADS)), dim3(RB_THREADS), 0, st, table_rows_in, n_in, K, nbr_out, n_out, tab_out, masks_out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// --------------------------------------------------------------------------- pair compaction (spconv-style lists)
// From the output-major table build, per kernel offset, the compact ascending list of (in,out) pairs and
// the pair counts — wavefront ballot + prefix popcount, order-preserving (ascending output row).
// Used for the Pairs statistic of the roofline model and by the weight-gradient kernel's work estimate.
__global__ __launch_bounds__(RB_THREADS) void k_pair_count(const int32_t* __restrict__ nbr, int64_t n_out, int K, int32_t* __restrict__ counts) {
  // grid: (blocks over rows, K)
  const int k = blockIdx.y;
  const int32_t* row = nbr + (int64_t)k * n_out;
  int local = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long m = __ballot(row[i] >= 0);
    if ((threadIdx.x & 63) == 0) local += __popcll(m);
  }
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(&counts[k], local);
}

extern "C" int sv_rulebook_pair_counts(const int32_t* nbr, int64_t n_out, int K, int32_t* counts, void* stream) {
  SV_CHECK_ARG(K > 0 && n_out >= 0 && counts, "rulebook_pair_counts: bad arguments");
  hipStream_t st = sv_stream(stream);
  SV_HIP(hipMemsetAsync(counts, 0, (size_t)K * 4, st));
  if (n_out == 0) return SV_OK;
  SV_CHECK_ARG(nbr, "rulebook_pair_counts: null pointer");
  dim3 grid(sv_grid_1d(n_out, RB_THREADS, 64), K);
  hipLaunchKernelGGL(k_pair_count, grid, dim3(RB_THREADS), 0, st, nbr, n_out, K, counts);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ===========================================================================================================================================
// A CHAIN of strided levels without a host read per level, and all tables of a network in three launches.
//
// The reference's backbones (spconv_backbone.py:141-157) run conv2 -> conv3 -> conv4 -> conv_out; spconv builds the indice pairs of each
// lazily and reads the number of output sites back to the host every time (four blocking reads per step here in round 3, each waiting for
// everything queued in front of it).  Two phases instead:
//
//  phase 1  sv_rulebook_chain_count: the OUTPUT SITE SETS of every level, end to end on the device.  Level l marks its output cells in its
//           occupancy bitmap from level l-1's site list (k_chain_mark_list: one lane per site, the list's length read from the device --
//           driving the marks from the bitmap below was built in round 3 and lost to divergence), counts the bitmap (k_chain_count: per-chunk
//           counts, in-chunk prefixes, per-block sums) and turns it into the ascending-key site list in a capacity buffer (k_chain_emit).
//           3 L launches, then ONE read for all L counts (+ the voxel count, whose device word feeds level 0's marks).
//  phase 2  sv_rulebook_batch: with the counts known every table has its exact size.  Three multi-job launches for the whole network:
//           SET   every level's sites into its dense cell -> row map (and: copy the sites into their exact-size tensor, return the
//                 bitmap words they set to zero),
//           QUERY every table -- submanifold (K, n), strided output-major (K, n_out) and input-major (K, n_in) -- as "row r, offset k ->
//                 one cell of a map": no atomics, no scattered stores, no -1 pre-fill, no inversion pass; each job also writes the
//                 row-major twin and the neighbour masks the conv plans consume,
//           CLEAR the maps.
// Same tables as sv_rulebook_sparse + sv_rulebook_invert_rows / sv_rulebook_subm_cellmap, bit for bit (tests/test_spconv.py).
// ===========================================================================================================================================
constexpr int CH_BLOCK = 256;          // chunks per workgroup of the count / emit passes (256 KiB of cells, 64 KiB of bitmap)
constexpr int CH_MAX_LEVELS = 8;
constexpr int CH_EMIT_SPLIT = 4;       // workgroups that share the emit of one block's non-empty chunks
static_assert(CH_BLOCK == RB_THREADS, "k_chain_emit scans one chunk count per thread");
static bool small_kernel_host(const ConvGeom& g) { return g.ksize[0] <= 3 && g.ksize[1] <= 3 && g.ksize[2] <= 3; }

template <bool FULL3>
__global__ __launch_bounds__(RB_THREADS) void k_chain_mark_list(const int4* __restrict__ coords, const int32_t* __restrict__ n_dev, int64_t n_host,
                                                                ConvGeom g, SvIndexView ix) {
  int64_t n = n_dev ? (int64_t)*n_dev : n_host;
  if (n > n_host) n = n_host;                          // with the count on the device n_host is the buffer's capacity
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = coords[i];
    if (coord_ok(c, g.batch, g.in_shape)) sparse_mark_one<FULL3>(c, g, ix, 0);
  }
}

// chunk_cnt[c], words[w].y (in-chunk prefix, only where the word is non-empty) and block_sums[b] = occupied cells of the block's CH_BLOCK chunks.
// 32 lanes per chunk; a thread requests the words of all its 32 chunks before it looks at the first (one load at a time the pass is a chain of
// 32 memory latencies: 22 us for a 47 MB bitmap).  Empty chunks (most of them) take no scan and no store: their count is zero already.
__global__ __launch_bounds__(RB_THREADS) void k_chain_count(SvIndexView ix, int64_t nchunks, int32_t* __restrict__ block_sums) {
  __shared__ int s_sum[RB_THREADS / 32];
  constexpr int PER = CH_BLOCK / (RB_THREADS / 32);        // chunks per 32-lane group
  const int lane = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int64_t c0 = (int64_t)blockIdx.x * CH_BLOCK;
  uint32_t bits[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int64_t c = c0 + sub + (RB_THREADS / 32) * i;
    bits[i] = c < nchunks ? ix.words[c * SV_CHUNK_WORDS + lane].x : 0u;
  }
  int mine = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const unsigned long long any = __ballot(bits[i] != 0u);
    if (!((any >> (32 * (sub & 1))) & 0xffffffffull)) continue;         // uniform over the 32-lane group
    const int64_t c = c0 + sub + (RB_THREADS / 32) * i;
    const int p = __popc(bits[i]);
    int incl = p;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int t = __shfl_up(incl, d, 32);
      if (lane >= d) incl += t;
    }
    if (bits[i]) ix.words[c * SV_CHUNK_WORDS + lane].y = (uint32_t)(incl - p);
    if (lane == 31) ix.chunk_cnt[c] = incl, mine += incl;
  }
  if (lane == 31) s_sum[sub] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int i = 0; i < RB_THREADS / 32; ++i) t += s_sum[i];
    block_sums[blockIdx.x] = t;
  }
}

// block b: rank base = sum of the blocks before it; the sites of its non-empty chunks -> sites[rank] (ascending key).  The next level's marks are a
// launch of their own (k_chain_mark_list with the count on the device): fused in here they were bound by the few blocks that hold most of a
// LiDAR scene's sites (186 us for level 1 of the bench batch).
// grid = (blocks, CH_EMIT_SPLIT): the same imbalance bounded this kernel too (a LiDAR scene's sites sit in a few z-slices: a quarter of the blocks
// hold nine tenths of them, 38 us standalone / 66 us inside the step per level), so workgroup (b, y) emits the y-th share of block b's non-empty
// chunks; the scans in front are repeated per share (256 chunk counts) and the shares of an empty block leave after them.
__global__ __launch_bounds__(RB_THREADS) void k_chain_emit(SvIndexView ix, int64_t nchunks, const int32_t* __restrict__ block_sums, int4* __restrict__ sites,
                                                           int64_t capacity, int32_t* __restrict__ num_out, ConvGeom gout /* shape of THIS level in out_shape */) {
  __shared__ int s_red[RB_THREADS / SV_WAVE], s_wtot[RB_THREADS / SV_WAVE], s_wne[RB_THREADS / SV_WAVE];
  __shared__ int s_cbase[CH_BLOCK];
  __shared__ uint16_t s_list[CH_BLOCK];
  const int tid = threadIdx.x, lane32 = tid & 31, sub = tid >> 5, b = blockIdx.x, lane = tid & 63, wid = tid >> 6;
  int before = 0;
  for (int j = tid; j < b; j += RB_THREADS) before += block_sums[j];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, SV_WAVE);
  // exclusive scan of the block's chunk counts (one chunk per thread: CH_BLOCK == RB_THREADS) and the list of its non-empty chunks
  const int64_t c_mine = (int64_t)b * CH_BLOCK + tid;
  const int cnt = c_mine < nchunks ? ix.chunk_cnt[c_mine] : 0;
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < SV_WAVE; d <<= 1) {
    const int t = __shfl_up(incl, d, SV_WAVE);
    if (lane >= d) incl += t;
  }
  const unsigned long long ne = __ballot(cnt > 0);
  if (lane == 0) s_red[wid] = before, s_wne[wid] = __popcll(ne);
  if (lane == 63) s_wtot[wid] = incl;
  __syncthreads();
  int base = 0, wbase = 0, total = 0, nebase = 0, n_ne = 0;
#pragma unroll
  for (int i = 0; i < RB_THREADS / SV_WAVE; ++i) {
    base += s_red[i];
    if (i < wid) wbase += s_wtot[i], nebase += s_wne[i];
    total += s_wtot[i], n_ne += s_wne[i];
  }
  s_cbase[tid] = base + wbase + incl - cnt;
  if (cnt > 0) s_list[nebase + __popcll(ne & ((1ull << lane) - 1ull))] = (uint16_t)tid;
  if (b == (int)gridDim.x - 1 && blockIdx.y == 0 && tid == 0) *num_out = base + total;
  __syncthreads();
  // this workgroup's share of the block's non-empty chunks
  const int q_lo = (int)((int64_t)n_ne * blockIdx.y / gridDim.y), q_hi = (int)((int64_t)n_ne * (blockIdx.y + 1) / gridDim.y);
  // emit: 32 lanes per non-empty chunk, a lane walks the set bits of its word.  The word's first key is decoded with divisions once; its 32
  // cells follow by carries (x-fastest key).  EM_B words are requested before the first is used.
  const int X = gout.out_shape[2], Y = gout.out_shape[1], Z = gout.out_shape[0];
  const bool narrow = ix.ncells < ((int64_t)1 << 31);
  constexpr int EM_B = 4, GROUPS = RB_THREADS / 32;
  for (int q0 = q_lo + sub; q0 < q_hi; q0 += GROUPS * EM_B) {
    uint2 wd[EM_B];
    int jj[EM_B];
#pragma unroll
    for (int u = 0; u < EM_B; ++u) {
      const int q = q0 + GROUPS * u;
      jj[u] = q < q_hi ? (int)s_list[q] : -1;
      wd[u] = jj[u] >= 0 ? ix.words[((int64_t)b * CH_BLOCK + jj[u]) * SV_CHUNK_WORDS + lane32] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < EM_B; ++u) {
      uint32_t bits = wd[u].x;
      if (!bits) continue;
      const int64_t w = ((int64_t)b * CH_BLOCK + jj[u]) * SV_CHUNK_WORDS + lane32;
      int64_t r = (int64_t)s_cbase[jj[u]] + wd[u].y;
      int bb, z0, y0, x0;
      if (narrow) {
        const uint32_t key = (uint32_t)w * 32u;
        const uint32_t q1 = key / (uint32_t)X;
        x0 = (int)(key - q1 * (uint32_t)X);
        const uint32_t q2 = q1 / (uint32_t)Y;
        y0 = (int)(q1 - q2 * (uint32_t)Y);
        bb = (int)(q2 / (uint32_t)Z);
        z0 = (int)(q2 - (uint32_t)bb * (uint32_t)Z);
      } else {
        const int64_t key = w * 32;
        const int64_t q1 = key / X, q2 = q1 / Y;
        x0 = (int)(key - q1 * X), y0 = (int)(q1 - q2 * Y), bb = (int)(q2 / Z), z0 = (int)(q2 - (int64_t)bb * Z);
      }
      while (bits) {
        const int bit = __ffs(bits) - 1;
        bits &= bits - 1u;
        int x = x0 + bit, y = y0, z = z0, bq = bb;
        while (x >= X) {
          x -= X;
          if (++y == Y) {
            y = 0;
            if (++z == Z) z = 0, ++bq;
          }
        }
        if (r < capacity) sites[r] = make_int4(bq, z, y, x);
        ++r;
      }
    }
  }
}

extern "C" size_t sv_rulebook_chain_scratch_bytes(int64_t max_ncells) {
  return align256((size_t)(sv_index_nchunks(max_ncells) / CH_BLOCK + 2) * sizeof(int32_t));
}

// geoms_host: n_levels x 15 int32 = {in_shape[3], ksize[3], stride[3], padding[3], dilation[3]} (level l's in_shape must be level l-1's output
// shape).  index_ws[l]: the persistent index of level l's OUTPUT grid (all zero on entry; left MARKED -- sv_rulebook_batch's SET jobs return the
// touched words to zero, or memset the workspace).  sites[l]: capacity caps[l] x int4; num_out: n_levels int32 on the device.
extern "C" int sv_rulebook_chain_count(const int32_t* coords0, int64_t n0, const int32_t* n0_dev, int batch, int n_levels, const int32_t* geoms_host,
                                       void* const* index_ws, int32_t* const* sites, const int64_t* caps, int32_t* num_out, void* scratch, void* stream) {
  SV_CHECK_ARG(batch > 0 && n_levels >= 1 && n_levels <= CH_MAX_LEVELS && n0 >= 0 && geoms_host && index_ws && sites && caps && num_out && scratch,
               "rulebook_chain_count: bad arguments (1 <= levels <= %d)", CH_MAX_LEVELS);
  SV_CHECK_ARG(coords0 || n0 == 0, "rulebook_chain_count: null coordinates");
  hipStream_t st = sv_stream(stream);
  ConvGeom g[CH_MAX_LEVELS];
  SvIndexView ix[CH_MAX_LEVELS];
  bool full3[CH_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l) {
    const int32_t* q = geoms_host + 15 * l;
    int rc = fill_geom(g[l], batch, q, q + 3, q + 6, q + 9, q + 12, false);
    if (rc) return rc;
    SV_CHECK_ARG(g[l].K <= RB_KMAX && small_kernel_host(g[l]), "rulebook_chain_count: kernels up to 3 per axis");
    if (l > 0)
      for (int d = 0; d < 3; ++d) SV_CHECK_ARG(g[l].in_shape[d] == g[l - 1].out_shape[d], "rulebook_chain_count: level %d does not take level %d's output shape", l, l - 1);
    SV_CHECK_ARG(index_ws[l] && sites[l] && caps[l] > 0, "rulebook_chain_count: null pointer at level %d", l);
    const int64_t ncells = (int64_t)batch * g[l].out_shape[0] * g[l].out_shape[1] * g[l].out_shape[2];
    ix[l] = sv_index_view(index_ws[l], ncells);
    full3[l] = g[l].ksize[0] == 3 && g[l].ksize[1] == 3 && g[l].ksize[2] == 3;
  }
  const int4* c4 = reinterpret_cast<const int4*>(coords0);
  if (n0 > 0) {
    const dim3 grid(sv_grid_1d(n0, RB_THREADS, 2048));
    if (full3[0]) hipLaunchKernelGGL(k_chain_mark_list<true>, grid, dim3(RB_THREADS), 0, st, c4, n0_dev, n0, g[0], ix[0]);
    else hipLaunchKernelGGL(k_chain_mark_list<false>, grid, dim3(RB_THREADS), 0, st, c4, n0_dev, n0, g[0], ix[0]);
  }
  int32_t* sums = reinterpret_cast<int32_t*>(scratch);
  for (int l = 0; l < n_levels; ++l) {
    const int64_t nchunks = sv_index_nchunks(ix[l].ncells);
    const int blocks = (int)((nchunks + CH_BLOCK - 1) / CH_BLOCK);
    int4* s4 = reinterpret_cast<int4*>(sites[l]);
    hipLaunchKernelGGL(k_chain_count, dim3(blocks), dim3(RB_THREADS), 0, st, ix[l], nchunks, sums);
    static const int emit_split = getenv("SEEVCN_EMIT_SPLIT") ? atoi(getenv("SEEVCN_EMIT_SPLIT")) : CH_EMIT_SPLIT;       // A/B: 1 = one workgroup per block
    hipLaunchKernelGGL(k_chain_emit, dim3(blocks, emit_split < 1 ? 1 : emit_split), dim3(RB_THREADS), 0, st, ix[l], nchunks, sums, s4, caps[l], num_out + l, g[l]);
    if (l + 1 < n_levels) {
      // the next level's marks from the sites just written; their number is num_out[l] on the device, the grid is sized for the capacity
      const dim3 grid(sv_grid_1d(caps[l], RB_THREADS, 2048));
      if (full3[l + 1]) hipLaunchKernelGGL(k_chain_mark_list<true>, grid, dim3(RB_THREADS), 0, st, s4, num_out + l, caps[l], g[l + 1], ix[l + 1]);
      else hipLaunchKernelGGL(k_chain_mark_list<false>, grid, dim3(RB_THREADS), 0, st, s4, num_out + l, caps[l], g[l + 1], ix[l + 1]);
    }
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------------------------------- phase 2: batch
constexpr int RBJ_MAX = 16;            // jobs per launch (kernel-argument block: 16 x 152 B)

struct SetJob {                        // every site of a level into the level's cell -> row map
  const int4* src;                     // the level's sites (n rows)
  int4* dst;                           // optional exact-size copy (the rulebook's out_indices)
  int32_t* map;                        // tiled cell -> row + 1 map of the level (nullable: copy / bitmap clean-up only)
  uint2* words;                        // optional: occupancy bitmap the sites were marked in (phase 1) -- their words and chunk counts return to zero
  int32_t* chunk_cnt;
  int64_t n;
  int shape[3];
  int batch;
  CellTiling t;
  int wg0;                             // first workgroup of the job in the launch
};
struct SetBatch { SetJob j[RBJ_MAX]; int n; };

template <typename B>
__device__ __forceinline__ int job_of_block(const B& b, int blk) {
  int j = 0;
#pragma unroll
  for (int q = 1; q < RBJ_MAX; ++q) j += (q < b.n && blk >= b.j[q].wg0) ? 1 : 0;
  return j;
}

__global__ __launch_bounds__(RB_THREADS) void k_batch_set(SetBatch b, int clear_only) {
  const int ji = job_of_block(b, blockIdx.x);
  const SetJob& J = b.j[ji];
  const int64_t i = (int64_t)(blockIdx.x - J.wg0) * RB_THREADS + threadIdx.x;
  if (i >= J.n) return;
  const int4 c = J.src[i];
  if (clear_only) {
    if (J.map && coord_ok(c, J.batch, J.shape)) J.map[tiled_cell(c.x, c.y, c.z, c.w, J.t)] = 0;
    return;
  }
  if (J.dst) J.dst[i] = c;
  if (!coord_ok(c, J.batch, J.shape)) return;
  if (J.map) J.map[tiled_cell(c.x, c.y, c.z, c.w, J.t)] = (int32_t)i + 1;
  if (J.words) {
    const int64_t key = lin_key(c.x, c.y, c.z, c.w, J.shape);
    J.words[key >> 5] = make_uint2(0u, 0u);
    J.chunk_cnt[key >> SV_CHUNK_SHIFT] = 0;
  }
}

// One table: row r (coordinate c) and kernel offset k = (kz, ky, kx) address the cell  t = (c * mul + add + k * step) / div  of the TARGET level
// (exact division, inside the target shape), whose map gives the source row.
//   submanifold:                 mul 1,      add -(ks/2)*dil, step +dil, div 1       (target = the level itself)
//   strided, output-major table: mul stride, add -pad,        step +dil, div 1       (rows = output sites, target = input level)
//   strided, input-major table:  mul 1,      add +pad,        step -dil, div stride  (rows = input sites,  target = output level)
struct QueryJob {
  const int4* rows;                    // (n) coordinates of the table's rows
  const int32_t* map;                  // target level's cell map
  int32_t* nbr;                        // (K, n) k-major table
  int32_t* tab;                        // (n, 32) row-major twin
  int32_t* masks;                      // (n)
  int64_t n;
  int rshape[3], tshape[3];            // shapes of the row level / target level
  int ksize[3], mul[3], add[3], step[3], div[3];
  int batch, K;
  CellTiling t;                        // tiling of the target map
  int wg0;
};
struct QueryBatch { QueryJob j[RBJ_MAX]; int n; };
static_assert(sizeof(QueryBatch) <= 3900 && sizeof(SetBatch) <= 3900, "kernel argument block");

template <bool FULL3>
__device__ __forceinline__ void query_row(const QueryJob& J, int64_t i) {
  const int4 c = J.rows[i];
  const bool ok = coord_ok(c, J.batch, J.rshape);
  const int cc[3] = {c.y, c.z, c.w};
  int cand[3][3];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const int dv = J.div[ax];
#pragma unroll
    for (int ka = 0; ka < 3; ++ka) {
      int o = -1;
      if (ka < J.ksize[ax]) {
        const int t = cc[ax] * J.mul[ax] + J.add[ax] + ka * J.step[ax];
        if (t >= 0) {
          if (dv == 1) o = t;
          else if (dv == 2) o = (t & 1) ? -1 : (t >> 1);
          else o = (t % dv) ? -1 : t / dv;
          if (o >= J.tshape[ax]) o = -1;
        }
      }
      cand[ax][ka] = o;
    }
  }
  const int kyx = J.ksize[1] * J.ksize[2];
  int32_t e[RB_ROW];
#pragma unroll
  for (int k = 0; k < RB_ROW; ++k) e[k] = -1;
#pragma unroll
  for (int k = 0; k < RB_KMAX; ++k) {
    if (k < J.K) {
      const int kz = FULL3 ? k / 9 : k / kyx, ky = FULL3 ? (k / 3) % 3 : (k - kz * kyx) / J.ksize[2], kx = FULL3 ? k % 3 : k - kz * kyx - ky * J.ksize[2];
      const int z = cand[0][kz % 3], y = cand[1][ky % 3], x = cand[2][kx % 3];
      if (ok && (z | y | x) >= 0) e[k] = J.map[tiled_cell(c.x, z, y, x, J.t)] - 1;
    }
  }
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < RB_KMAX; ++k) {
    if (k < J.K) {
      J.nbr[(int64_t)k * J.n + i] = e[k];
      m |= e[k] >= 0 ? (1u << k) : 0u;
    }
  }
  J.masks[i] = (int32_t)m;
  rb_i32x4* dst = reinterpret_cast<rb_i32x4*>(J.tab + i * RB_ROW);
#pragma unroll
  for (int q = 0; q < RB_ROW / 4; ++q) dst[q] = (rb_i32x4){e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]};
}

__global__ __launch_bounds__(RB_THREADS) void k_batch_query(QueryBatch b) {
  const int ji = job_of_block(b, blockIdx.x);
  const QueryJob& J = b.j[ji];
  const int64_t i = (int64_t)(blockIdx.x - J.wg0) * RB_THREADS + threadIdx.x;
  if (i >= J.n) return;
  if (J.ksize[0] == 3 && J.ksize[1] == 3 && J.ksize[2] == 3) query_row<true>(J, i);      // job-uniform, hence workgroup-uniform
  else query_row<false>(J, i);
}

// jobs_host: n_jobs rows of 32 int64.  row[0] = kind:
//   1 SET:    [1] src sites, [2] dst sites or 0, [3] cell map or 0, [4] index workspace to clean or 0, [5] n, [6..8] shape (Z, Y, X), [9] batch
//   2 QUERY:  [1] row sites, [2] target cell map, [3] nbr (K, n), [4] tab (n, 32), [5] masks (n), [6] n, [7..9] row-level shape, [10..12] target shape,
//             [13..15] ksize, [16..18] mul, [19..21] add, [22..24] step, [25..27] div, [28] batch
// Order of execution: every SET, then every QUERY, then the maps of every SET with a map are returned to zero (three launches; more than RBJ_MAX
// jobs of a kind are run in groups).  A SET that cleans an index takes the index's level shape from [6..8].
extern "C" int sv_rulebook_batch(const int64_t* jobs_host, int n_jobs, void* stream) {
  SV_CHECK_ARG(n_jobs >= 0 && (jobs_host || n_jobs == 0), "rulebook_batch: bad arguments");
  hipStream_t st = sv_stream(stream);
  SetBatch sb;
  QueryBatch qb;
  auto flush_set = [&](int clear_only, int wgs) {
    if (sb.n > 0 && wgs > 0) hipLaunchKernelGGL(k_batch_set, dim3(wgs), dim3(RB_THREADS), 0, st, sb, clear_only);
  };
  for (int pass = 0; pass < 3; ++pass) {              // 0: SET, 1: QUERY, 2: CLEAR
    sb.n = 0, qb.n = 0;
    int wgs = 0;
    for (int q = 0; q < n_jobs; ++q) {
      const int64_t* r = jobs_host + 32 * q;
      if (r[0] == 1 && pass != 1) {
        const int64_t n = r[5];
        SV_CHECK_ARG(n >= 0 && r[9] > 0, "rulebook_batch: SET job %d: bad sizes", q);
        if (n == 0 || (pass == 2 && !r[3])) continue;
        SV_CHECK_ARG(r[1], "rulebook_batch: SET job %d: null sites", q);
        SetJob& J = sb.j[sb.n];
        J.src = reinterpret_cast<const int4*>(r[1]), J.dst = reinterpret_cast<int4*>(r[2]), J.map = reinterpret_cast<int32_t*>(r[3]);
        J.n = n, J.batch = (int)r[9];
        for (int d = 0; d < 3; ++d) J.shape[d] = (int)r[6 + d];
        J.t = cell_tiling(J.shape);
        J.words = nullptr, J.chunk_cnt = nullptr;
        if (r[4]) {
          SvIndexView ix = sv_index_view(reinterpret_cast<void*>(r[4]), (int64_t)J.batch * J.shape[0] * J.shape[1] * J.shape[2]);
          J.words = ix.words, J.chunk_cnt = ix.chunk_cnt;
        }
        J.wg0 = wgs;
        wgs += sv_div_up(n, RB_THREADS);
        if (++sb.n == RBJ_MAX) flush_set(pass == 2, wgs), sb.n = 0, wgs = 0;
      } else if (r[0] == 2 && pass == 1) {
        const int64_t n = r[6];
        SV_CHECK_ARG(n >= 0 && r[28] > 0, "rulebook_batch: QUERY job %d: bad sizes", q);
        if (n == 0) continue;
        SV_CHECK_ARG(r[1] && r[2] && r[3] && r[4] && r[5], "rulebook_batch: QUERY job %d: null pointer", q);
        QueryJob& J = qb.j[qb.n];
        J.rows = reinterpret_cast<const int4*>(r[1]), J.map = reinterpret_cast<const int32_t*>(r[2]), J.nbr = reinterpret_cast<int32_t*>(r[3]);
        J.tab = reinterpret_cast<int32_t*>(r[4]), J.masks = reinterpret_cast<int32_t*>(r[5]), J.n = n, J.batch = (int)r[28];
        J.K = 1;
        for (int d = 0; d < 3; ++d) {
          J.rshape[d] = (int)r[7 + d], J.tshape[d] = (int)r[10 + d], J.ksize[d] = (int)r[13 + d], J.mul[d] = (int)r[16 + d], J.add[d] = (int)r[19 + d];
          J.step[d] = (int)r[22 + d], J.div[d] = (int)r[25 + d];
          SV_CHECK_ARG(J.ksize[d] >= 1 && J.ksize[d] <= 3 && J.div[d] >= 1 && J.rshape[d] > 0 && J.tshape[d] > 0, "rulebook_batch: QUERY job %d: bad geometry", q);
          J.K *= J.ksize[d];
        }
        J.t = cell_tiling(J.tshape);
        J.wg0 = wgs;
        wgs += sv_div_up(n, RB_THREADS);
        if (++qb.n == RBJ_MAX) {
          hipLaunchKernelGGL(k_batch_query, dim3(wgs), dim3(RB_THREADS), 0, st, qb);
          qb.n = 0, wgs = 0;
        }
      } else if (r[0] != 1 && r[0] != 2) {
        SV_CHECK_ARG(false, "rulebook_batch: unknown job kind %lld at position %d", (long long)r[0], q);
      }
    }
    if (pass == 1) {
      if (qb.n > 0) hipLaunchKernelGGL(k_batch_query, dim3(wgs), dim3(RB_THREADS), 0, st, qb);
    } else {
      flush_set(pass == 2, wgs);
    }
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}
