// Coordinate index: chunk-count exclusive scan + library-wide error state.
#include <stdarg.h>
#include <string.h>

#include "common.h"

// ------------------------------------------------------------------ error state (per host thread)
static thread_local char g_err[512] = "";

void sv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sv_last_error(void) { return g_err; }
extern "C" int sv_abi_version(void) { return SV_ABI_VERSION; }

extern "C" size_t sv_index_persistent_bytes(int64_t ncells) {
  const int64_t nchunks = sv_index_nchunks(ncells);
  // words (padded to whole chunks) + chunk_cnt + chunk_base
  return (size_t)nchunks * SV_CHUNK_WORDS * sizeof(uint2) + (size_t)nchunks * 2 * sizeof(int32_t);
}

// ------------------------------------------------------------------ exclusive scan of int32
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 4096 items per workgroup

__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = threadIdx.x & (SV_WAVE - 1);
#pragma unroll
  for (int d = 1; d < SV_WAVE; d <<= 1) {
    const int t = __shfl_up(v, d, SV_WAVE);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan of one value per thread across the workgroup; returns exclusive prefix, *total = sum
template <int THREADS>
__device__ __forceinline__ int block_excl_scan(int v, int* total) {
  __shared__ int wsum[THREADS / SV_WAVE];
  const int lane = threadIdx.x & (SV_WAVE - 1);
  const int wid = threadIdx.x / SV_WAVE;
  const int incl = wave_incl_scan(v);
  if (lane == SV_WAVE - 1) wsum[wid] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < THREADS / SV_WAVE; ++i) {
    const int s = wsum[i];
    if (i < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const int32_t* __restrict__ in, int64_t n,
                                                                  int32_t* __restrict__ block_sums) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int s = 0;
  if (base + SCAN_ITEMS <= n) {
    const int4* p = reinterpret_cast<const int4*>(in + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; ++i) {
      const int4 v = p[i];
      s += v.x + v.y + v.z + v.w;
    }
  } else {
    for (int i = 0; i < SCAN_ITEMS; ++i)
      if (base + i < n) s += in[base + i];
  }
  int tot;
  block_excl_scan<SCAN_THREADS>(s, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// Every workgroup adds up the sums of the blocks before it by itself (a few hundred values, coalesced) instead of reading them from a
// scan made by a one-workgroup kernel in between: two launches per scan instead of three.  The last block also writes the grand total.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_downsweep(const int32_t* __restrict__ in, int64_t n,
                                                                 const int32_t* __restrict__ block_sums,
                                                                 int32_t* __restrict__ out, int32_t* __restrict__ total_out) {
  int before = 0;
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += SCAN_THREADS) before += block_sums[j];
  int block_off;
  block_excl_scan<SCAN_THREADS>(before, &block_off);
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  const bool full = base + SCAN_ITEMS <= n;
  if (full) {
    const int4* p = reinterpret_cast<const int4*>(in + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; ++i) {
      const int4 q = p[i];
      v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) v[i] = (base + i < n) ? in[base + i] : 0;
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { const int t = v[i]; v[i] = s; s += t; }
  int tot;
  const int off = block_excl_scan<SCAN_THREADS>(s, &tot) + block_off;
  if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = block_off + tot;
  if (full) {
    int4* q = reinterpret_cast<int4*>(out + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; ++i)
      q[i] = make_int4(v[4 * i] + off, v[4 * i + 1] + off, v[4 * i + 2] + off, v[4 * i + 3] + off);
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
      if (base + i < n) out[base + i] = v[i] + off;
  }
}

size_t sv_index_scan_tmp_bytes(int64_t ncells) {
  const int64_t nchunks = sv_index_nchunks(ncells);
  const int64_t nblocks = (nchunks + SCAN_TILE - 1) / SCAN_TILE;
  return (size_t)((nblocks + 63) / 64 * 64) * sizeof(int32_t);
}

extern "C" size_t sv_index_scratch_bytes(int64_t ncells) { return sv_index_scan_tmp_bytes(ncells); }

// chunk_cnt -> chunk_base (exclusive), *total_out = number of distinct cells
int sv_index_scan_launch(const SvIndexView& ix, int32_t* total_out, void* scan_tmp, hipStream_t st) {
  const int64_t nchunks = sv_index_nchunks(ix.ncells);
  const int nblocks = (int)((nchunks + SCAN_TILE - 1) / SCAN_TILE);
  int32_t* sums = reinterpret_cast<int32_t*>(scan_tmp);
  hipLaunchKernelGGL(k_scan_block_sums, dim3(nblocks), dim3(SCAN_THREADS), 0, st, ix.chunk_cnt, nchunks, sums);
  hipLaunchKernelGGL(k_scan_downsweep, dim3(nblocks), dim3(SCAN_THREADS), 0, st, ix.chunk_cnt, nchunks, sums,
                     ix.chunk_base, total_out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
