// Shared helpers for the seevcn HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/seevcn_hip.h"

#define SV_WAVE 64

// Error reporting: entry points return 0 on success or a non-zero code and never exit()
// (the reference's launchers call exit(-1), e.g. ops/pointnet2/pointnet2_stack/src/ball_query_gpu.cu:85-89).
void sv_set_error(const char* fmt, ...);

#define SV_CHECK_ARG(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      sv_set_error(__VA_ARGS__);         \
      return SV_ERR_ARG;                 \
    }                                    \
  } while (0)

#define SV_HIP(call)                                                               \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      sv_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return SV_ERR_HIP;                                                           \
    }                                                                              \
  } while (0)

#define SV_LAUNCH_CHECK() SV_HIP(hipGetLastError())

static inline hipStream_t sv_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int sv_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Grid for a grid-stride 1-D kernel: enough workgroups to fill 256 CUs x 8, never more than the work.
static inline int sv_grid_1d(int64_t n, int block, int max_blocks = 256 * 8) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------
// Coordinate index ("rank dictionary") over a dense cell grid, kept in a persistent zeroed workspace.
//   words[w]      = {bits of cells 32w..32w+31, number of set cells in all words < w}
//   chunk_cnt[c]  = number of set cells in chunk c (SV_CHUNK_WORDS words), chunk_base = its exclusive scan
// rank(key) = words[key>>5].y + popc(bits below key) = position of key in ascending key order.
// ---------------------------------------------------------------------------------------------
#define SV_CHUNK_WORDS 32
#define SV_CHUNK_SHIFT 10  // log2(32 words * 32 cells)

struct SvIndexView {
  uint2* words;
  int32_t* chunk_cnt;
  int32_t* chunk_base;
  int64_t ncells;
};

static inline int64_t sv_index_nwords(int64_t ncells) { return (ncells + 31) / 32; }
static inline int64_t sv_index_nchunks(int64_t ncells) { return (ncells + 1023) / 1024; }

static inline SvIndexView sv_index_view(void* ws, int64_t ncells) {
  // layout: words (8 B each, padded to a whole chunk) | chunk_cnt | chunk_base
  SvIndexView v;
  int64_t nchunks = sv_index_nchunks(ncells);
  v.words = reinterpret_cast<uint2*>(ws);
  v.chunk_cnt = reinterpret_cast<int32_t*>(v.words + nchunks * SV_CHUNK_WORDS);
  v.chunk_base = v.chunk_cnt + nchunks;
  v.ncells = ncells;
  return v;
}

#ifdef __HIPCC__
// mark `key` present; bumps the chunk's distinct-cell count the first time the cell is seen
__device__ __forceinline__ void sv_index_mark(const SvIndexView& ix, int64_t key) {
  const int64_t w = key >> 5;
  const uint32_t bit = 1u << (key & 31);
  const uint32_t old = atomicOr(&ix.words[w].x, bit);
  if (!(old & bit)) atomicAdd(&ix.chunk_cnt[key >> SV_CHUNK_SHIFT], 1);
}

// after the chunk scan: rank of a key known to be present, computed from the chunk base and the
// popcounts of the words before it in its chunk (does not need words[].y)
__device__ __forceinline__ int32_t sv_index_rank_slow(const SvIndexView& ix, int64_t key, uint32_t* prefix_out) {
  const int64_t w = key >> 5;
  const int64_t w0 = (w / SV_CHUNK_WORDS) * SV_CHUNK_WORDS;
  int32_t p = ix.chunk_base[key >> SV_CHUNK_SHIFT];
  for (int64_t j = w0; j < w; ++j) p += __popc(ix.words[j].x);
  if (prefix_out) *prefix_out = (uint32_t)p;
  const uint32_t bits = ix.words[w].x;
  return p + __popc(bits & ((1u << (key & 31)) - 1u));
}

// after prefixes are stored: rank of key or -1 when absent (one 8-byte read)
__device__ __forceinline__ int32_t sv_index_lookup(const SvIndexView& ix, int64_t key) {
  const uint2 wd = ix.words[key >> 5];
  const uint32_t bit = 1u << (key & 31);
  if (!(wd.x & bit)) return -1;
  return (int32_t)wd.y + __popc(wd.x & (bit - 1u));
}
#endif

// host-side launch helpers implemented in coord_index.hip
int sv_index_scan_launch(const SvIndexView& ix, int32_t* total_out, void* scan_tmp, hipStream_t st);
size_t sv_index_scan_tmp_bytes(int64_t ncells);
