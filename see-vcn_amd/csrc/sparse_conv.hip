// Sparse 3-D convolution: output-stationary gather-GEMM over the output-major rulebook.
//
//   Y[o][n] = epilogue( sum_k sum_c X[nbr[k][o]][c] * Wt[k][n][c] )       (rows with nbr < 0 contribute 0)
//
// One kernel serves three uses (the caller picks the table and the weight view):
//   forward        X = features,  nbr = output-major table,            Wt[k][n][c] = W[k][c_in=c][c_out=n]
//   backward-data  X = grad_out,  nbr = input-major table (nbr_in),    Wt[k][n][c] = W[k][c_in=n][c_out=c]
// and a second kernel reduces the weight gradient dW[k][c][n] = sum_o X[nbr[k][o]][c] * dY[o][n].
//
// Every output row is produced exactly once, in registers, with a fixed summation order (k ascending):
// no atomics, bitwise reproducible.  The dense per-voxel products run on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32, exact fp32) — 16-row tiles so that a (tile, offset) pair with no neighbour is skipped.
// Algorithmic traffic per layer: 4*(N_in*C_in + N_out*C_out) + 4*K*C_in*C_out + 4*K*N_out (table) bytes.
//
// Replaces the third-party spconv kernels behind SubMConv3d / SparseConv3d
// (call sites: detector3d/pcdet/models/backbones_3d/spconv_backbone.py:8-27,77-117).
#include <stdlib.h>

#include "common.h"


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SC_THREADS = 256;
constexpr int SC_ROWS_PER_WAVE = 32;  // two 16-row MFMA tiles
constexpr int SC_ROWS_PER_BLOCK = SC_ROWS_PER_WAVE * (SC_THREADS / 64);
constexpr int SC_KSLICE = 64;         // contraction channels staged in LDS at a time

struct ConvArgs {
  const float* X;         // (n_src, Kd)
  const int32_t* nbr;     // (K, n_rows)
  const float* Wt;        // (K, Nc, Kd)
  float* Y;               // (n_rows, Nc)
  const float* bias;      // (Nc) or null        : y += bias
  const float* scale;     // (Nc) or null        : y = y*scale + shift   (folded eval-mode BatchNorm)
  const float* shift;     // (Nc) or null
  const float* residual;  // (n_rows, Nc) or null: y += residual (after scale/shift, before relu)
  int relu;
  int64_t n_rows;
  int K, Kd, Nc;
  const int32_t* tile_order;  // [wave * 4 + slot] -> 16-row tile (sv_conv_tile_order) or null: tiles by position
  const int32_t* row_perm;    // table column p produces output row row_perm[p] (sv_conv_group_rows), or null: p itself
  int k_flip;                 // read table row K-1-k for offset k (a submanifold table serving its own data gradient)
};

__device__ __forceinline__ float conv_epilogue(float v, int col, int64_t row, const ConvArgs& a) {
  if (a.bias) v += a.bias[col];
  if (a.scale) v = v * a.scale[col] + a.shift[col];
  if (a.residual) v += a.residual[row * a.Nc + col];
  if (a.relu) v = fmaxf(v, 0.f);
  return v;
}

// NT = Nc/16 column tiles held in registers. LDS holds a slice of Wt[k] as [Nc][min(Kd,64)+4].
template <int NT>
__global__ __launch_bounds__(SC_THREADS) void k_spconv_mfma(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Ws[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const int pitch = min(a.Kd, SC_KSLICE) + 4;
  const int64_t row0 = (int64_t)blockIdx.x * SC_ROWS_PER_BLOCK + wid * SC_ROWS_PER_WAVE;

  f32x4 acc[2][NT];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k = 0; k < a.K; ++k) {
    // gather indices of this wave's 2x16 rows for offset k (coalesced 64 B reads)
    int32_t j[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int64_t r = row0 + g * 16 + li;
      j[g] = r < a.n_rows ? a.nbr[(int64_t)k * a.n_rows + r] : -1;
    }
    const bool any0 = __ballot(j[0] >= 0) != 0ull, any1 = __ballot(j[1] >= 0) != 0ull;
    const unsigned long long blk_any = __syncthreads_or(any0 || any1);
    if (!blk_any) continue;                // nobody in the workgroup needs W[k] (uniform across the block)
    // stage Wt[k] -> LDS in contraction slices of <= SC_KSLICE channels (bounds LDS at Nc*(SC_KSLICE+4)*4 bytes)
    for (int ks = 0; ks < a.Kd; ks += SC_KSLICE) {
      const int kw = min(SC_KSLICE, a.Kd - ks);      // slice width (multiple of 16)
      const int wq = kw / 4;                          // float4 per weight row in the slice
      const float* src = a.Wt + (int64_t)k * a.Nc * a.Kd + ks;
      for (int e = tid; e < a.Nc * wq; e += SC_THREADS) {
        const int n = e / wq, c4 = e - n * wq;
        *reinterpret_cast<float4*>(&Ws[n * pitch + c4 * 4]) = *reinterpret_cast<const float4*>(src + (int64_t)n * a.Kd + c4 * 4);
      }
      __syncthreads();
      if (any0 || any1) {
        for (int q = 0; q < kw / 16; ++q) {
          float4 av[2];
#pragma unroll
          for (int g = 0; g < 2; ++g)
            av[g] = j[g] >= 0 ? *reinterpret_cast<const float4*>(a.X + (int64_t)j[g] * a.Kd + ks + q * 16 + kk * 4) : make_float4(0, 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float4 b = *reinterpret_cast<const float4*>(&Ws[(t * 16 + li) * pitch + q * 16 + kk * 4]);
            if (any0) {
              acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].x, b.x, acc[0][t], 0, 0, 0);
              acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].y, b.y, acc[0][t], 0, 0, 0);
              acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].z, b.z, acc[0][t], 0, 0, 0);
              acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].w, b.w, acc[0][t], 0, 0, 0);
            }
            if (any1) {
              acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].x, b.x, acc[1][t], 0, 0, 0);
              acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].y, b.y, acc[1][t], 0, 0, 0);
              acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].z, b.z, acc[1][t], 0, 0, 0);
              acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].w, b.w, acc[1][t], 0, 0, 0);
            }
          }
        }
      }
      __syncthreads();                     // Ws is overwritten by the next slice / offset
    }
  }
  // D layout (16x16): col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + g * 16 + kk * 4 + r;
        const int col = t * 16 + li;
        if (row < a.n_rows) a.Y[row * a.Nc + col] = conv_epilogue(acc[g][t][r], col, row, a);
      }
}

// ------------------------------------------------------------------------------------------------
// Register-stationary, barrier-free variant (the default MFMA path).
//   * a wave owns RS_G 16-row output tiles x all Nc columns, accumulators in registers;
//   * no LDS, no workgroup barriers: both MFMA operands are loaded straight into registers — the gathered
//     source rows (A) and the 16-column weight slabs (B, shared by every wave, served by L1/L2);
//   * operands of step (k, q) + 1 are requested before step (k, q) runs on the matrix core, the neighbour
//     indices of the next active offset are fetched one offset ahead;
//   * offsets with no neighbour for any of the wave's rows are skipped entirely (64-bit activity mask from a
//     ballot pre-pass), row tiles with no neighbour skip their MFMAs;
//   * the wave's tiles are taken from RS_G distant parts of the (key-sorted) row range: neighbour density is
//     spatially correlated, striding gives every wave a mix of dense and sparse regions (measured -8 % time).
// Summation order per output element is fixed (k ascending, channels ascending) -> bitwise reproducible.
// Measured (MI355X, 64->64 submanifold layer, 134 580 rows, 1.17 M pairs): 250 us = 38 TFLOP/s algorithmic;
// matrix-core busy 38 % — waves spend their time in issue stalls, see DESIGN.md "sparse conv: what limits it".
// Two LDS-DMA restructurings (global_load_lds row gathers into a ring, offset-major weight slabs) were built, verified
// bit-compatible with the tests and measured slower (304 us with columns split over a workgroup's waves and a barrier per
// position; 266 us wave-private with the next slab prefetched) — they are in the git history (commit "Experimental
// wave-private LDS-DMA sparse conv kernel"), the findings in DESIGN.md.
// ------------------------------------------------------------------------------------------------
template <int NT, int KQ, int RS_G>
__global__ __launch_bounds__(256) void k_spconv_rs(ConvArgs a) {
  constexpr int Kd = KQ * 16, Nc = NT * 16;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const int64_t n_tiles = (a.n_rows + 15) / 16;
  const int64_t n_waves = (n_tiles + RS_G - 1) / RS_G;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wid;
  if (wave_id >= n_waves) return;
  auto tile_row0 = [&](int g) { return (wave_id + (int64_t)g * n_waves) * 16; };

  // activity mask of the kernel offsets for this wave's rows
  unsigned long long active = 0ull;
  {
    // lane -> (tile lane>>4, row lane&15); with RS_G < 4 the upper tiles alias tile 0 (harmless for an OR)
    const int64_t r = tile_row0((lane >> 4) % RS_G) + li;
    for (int k = 0; k < a.K; ++k) {
      const int32_t j = r < a.n_rows ? a.nbr[(int64_t)k * a.n_rows + r] : -1;
      if (__ballot(j >= 0)) active |= 1ull << k;
    }
  }

  f32x4 acc[RS_G][NT];
#pragma unroll
  for (int g = 0; g < RS_G; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto load_j = [&](int k, int32_t (&j)[RS_G]) {
#pragma unroll
    for (int g = 0; g < RS_G; ++g) {
      const int64_t r = tile_row0(g) + li;
      j[g] = r < a.n_rows ? a.nbr[(int64_t)k * a.n_rows + r] : -1;
    }
  };
  auto load_ab = [&](int k, int q, const int32_t (&j)[RS_G], float4 (&A)[RS_G], float4 (&B)[NT]) {
#pragma unroll
    for (int g = 0; g < RS_G; ++g)
      A[g] = j[g] >= 0 ? *reinterpret_cast<const float4*>(a.X + (int64_t)j[g] * Kd + q * 16 + kk * 4) : make_float4(0, 0, 0, 0);
    const float* w = a.Wt + ((int64_t)k * Nc + li) * Kd + q * 16 + kk * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) B[t] = *reinterpret_cast<const float4*>(w + (int64_t)t * 16 * Kd);
  };

  if (active) {
    int k = __ffsll((long long)active) - 1;
    active &= active - 1;
    int32_t jc[RS_G], jn[RS_G];
    float4 Ac[RS_G], Bc[NT], An[RS_G], Bn[NT];
    load_j(k, jc);
    load_ab(k, 0, jc, Ac, Bc);
    while (true) {
      const int kn = active ? __ffsll((long long)active) - 1 : -1;
      if (kn >= 0) load_j(kn, jn);
      bool anyg[RS_G];
#pragma unroll
      for (int g = 0; g < RS_G; ++g) anyg[g] = __ballot(jc[g] >= 0) != 0ull;
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        // request the next step's operands before running this step's MFMAs
        if (q + 1 < KQ) load_ab(k, q + 1, jc, An, Bn);
        else if (kn >= 0) load_ab(kn, 0, jn, An, Bn);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int g = 0; g < RS_G; ++g)
            if (anyg[g]) {
              acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[g].x, Bc[t].x, acc[g][t], 0, 0, 0);
              acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[g].y, Bc[t].y, acc[g][t], 0, 0, 0);
              acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[g].z, Bc[t].z, acc[g][t], 0, 0, 0);
              acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[g].w, Bc[t].w, acc[g][t], 0, 0, 0);
            }
        if (q + 1 < KQ || kn >= 0) {
#pragma unroll
          for (int g = 0; g < RS_G; ++g) Ac[g] = An[g];
#pragma unroll
          for (int t = 0; t < NT; ++t) Bc[t] = Bn[t];
        }
      }
      if (kn < 0) break;
      k = kn;
      active &= active - 1;
#pragma unroll
      for (int g = 0; g < RS_G; ++g) jc[g] = jn[g];
    }
  }
  // D layout (16x16): col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int g = 0; g < RS_G; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = tile_row0(g) + kk * 4 + r;
        const int col = t * 16 + li;
        if (row < a.n_rows) a.Y[row * Nc + col] = conv_epilogue(acc[g][t][r], col, row, a);
      }
}

// ------------------------------------------------------------------------------------------------
// Three-stage variant of k_spconv_rs.  PMC on k_spconv_rs: matrix core busy 38 %, waves waiting for operands that were requested
// only one 64-MFMA step (~2000 cycles) earlier, less than the gather latency under load; hipcc additionally sinks its own
// prefetch loads towards their first use.  Here every operand load of the loop is an inline-asm buffer_load_dwordx4 (hipcc can
// neither move it nor wait for it), issued TWO steps ahead into a 3-deep register ring, and retired with a counted
// s_waitcnt vmcnt(2 x loads-per-step) that names the stage's registers ("+v", form (ii) of cdna_hip_programming.md 5.7).
// Every step issues exactly RS_G + NT loads: rows without a neighbour use an out-of-range buffer offset (the range check
// returns zeros without a memory access), steps past the end issue out-of-range dummies.  The wave's neighbour indices are parked
// in LDS once, so the loop contains no compiler-visible VMEM load.  Same ownership, skipping and summation order as k_spconv_rs.
// ------------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 make_srd(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));      // stride 0
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ f32x4 buf_load_b128(i32x4 srd, uint32_t voff) {
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(srd) : "memory");
  return v;
}

constexpr int RS3_KMAX = 27;

// weights re-laid in MFMA fragment order, one contiguous KiB per (offset, 16-channel step, column tile): the B-operand load of
// a wave then touches 8 whole cache lines instead of 16 half lines at 16 different rows.  Rewritten by every call (<= 442 KB);
// calls on one device are serialised on one stream ("Error behaviour and streams").
__device__ __attribute__((aligned(256))) float g_wfrag[RS3_KMAX * 64 * 64];

struct WStride {
  int64_t k, n, c;     // element strides of the (K, Nc, Kd) weight view
};
__global__ __launch_bounds__(256) void k_weight_fragments(const float* __restrict__ wt, WStride ws, int K, int Nc, int Kd, float* __restrict__ wf) {
  const int total = K * Nc * Kd / 4;                      // float4 units
  const int KQ = Kd / 16, NT = Nc / 16;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63, t = (i >> 6) % NT, q = ((i >> 6) / NT) % KQ, k = (i >> 6) / (NT * KQ);
    const int li = lane & 15, kk = lane >> 4;
    const float* src = wt + k * ws.k + (t * 16 + li) * ws.n + (q * 16 + kk * 4) * ws.c;
    float4 v;
    if (ws.c == 1 && (((uintptr_t)src) & 15) == 0) v = *reinterpret_cast<const float4*>(src);
    else v = make_float4(src[0], src[ws.c], src[2 * ws.c], src[3 * ws.c]);
    reinterpret_cast<float4*>(wf)[i] = v;
  }
}
// any weight view -> contiguous (K, Nc, Kd) for the kernels that read the weights in place
constexpr int64_t WPACK_FLOATS = 27 * 128 * 128;
__device__ __attribute__((aligned(256))) float g_wpack[WPACK_FLOATS];
__global__ __launch_bounds__(256) void k_weight_pack(const float* __restrict__ wt, WStride ws, int K, int Nc, int Kd, float* __restrict__ out) {
  const int64_t total = (int64_t)K * Nc * Kd;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Kd), n = (int)((i / Kd) % Nc), k = (int)(i / ((int64_t)Kd * Nc));
    out[i] = wt[k * ws.k + n * ws.n + c * ws.c];
  }
}

// Neighbour mask of every row of a table: bit k set iff nbr[k][row] >= 0 (K <= 32).  Rows with equal masks make 16-row tiles whose
// every executed (tile, offset) step is useful; tiles of consecutive rows waste 40-80 % of them.
__global__ __launch_bounds__(256) void k_row_masks(const int32_t* __restrict__ nbr, int64_t n_rows, int K, int32_t* __restrict__ masks) {
  for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n_rows; row += (int64_t)gridDim.x * 256) {
    unsigned m = 0;
    for (int k0 = 0; k0 < K; k0 += 9) {
      int32_t j[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) j[u] = k0 + u < K ? nbr[(int64_t)(k0 + u) * n_rows + row] : -1;
#pragma unroll
      for (int u = 0; u < 9; ++u) m |= j[u] >= 0 ? (1u << (k0 + u)) : 0u;
    }
    masks[row] = (int32_t)m;
  }
}

extern "C" int sv_conv_row_masks(const int32_t* nbr, int64_t n_rows, int K, int32_t* masks, void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && K > 0 && K <= 31, "sv_conv_row_masks: 1 <= K <= 31 (got %d)", K);
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(nbr && masks, "sv_conv_row_masks: null pointer");
  hipLaunchKernelGGL(k_row_masks, dim3(sv_grid_1d(n_rows, 256)), dim3(256), 0, sv_stream(stream), nbr, n_rows, K, masks);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// Regrouping of a table's columns by neighbour mask without a sort: a counting sort over GR_BUCKETS classes of the mask
// (group_key).  Three launches: masks + class histogram, an exclusive scan over the classes, and the placement row_perm[p] = row.
// The table itself is NOT rewritten: the conv kernel reads its 16 x 27 entries per tile through row_perm (a 2 us start-up per
// wave instead of a 50 MB pass per table).  Histogram and cursors are bumped once per (wave, class) -- lanes with equal keys are
// found with ballots -- so the hot classes (one mask covers ~20 % of the rows) do not serialise on one address.
// `hist` is persistent and all-zero between calls.
constexpr int GR_BUCKETS = 4096;
__device__ __forceinline__ int group_key(unsigned mask) {
  // equal masks -> equal class; classes ordered by the number of active offsets first (neighbouring tiles then cost the same and
  // mixed tiles at class borders waste little), a 7-bit hash of the mask inside one count
  return (__popc(mask) << 7) | (int)((mask * 2654435761u) >> 25);
}
struct GroupArgs {
  const int32_t* nbr;
  int64_t n_rows;
  int K;
  int32_t* masks;     // out (n_rows)
  int32_t* hist;      // [0..B): class counts (zero on entry, zeroed again by the scan); [B..2B): class starts; [2B..3B): cursors
  int32_t* perm;      // out (n_rows)
};

// "for every distinct key among the live lanes": this lane's rank inside its key group, the group's size and its first lane
__device__ __forceinline__ void wave_key_groups(int key, bool live, int& rank, int& size, int& first_lane) {
  unsigned long long todo = __ballot(live);
  const int lane = threadIdx.x & 63;
  rank = 0, size = 0, first_lane = lane;
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const int k0 = __shfl(key, first);
    const unsigned long long same = __ballot(live && key == k0);
    if (live && key == k0) {
      rank = __popcll(same & ((1ull << lane) - 1));
      size = __popcll(same);
      first_lane = first;
    }
    todo &= ~same;
  }
}

// 1024 rows per workgroup, one per thread.  Class counts go wave -> LDS (one atomic per (wave, class)) -> global (one atomic per
// (workgroup, class) that occurs): the hottest class sees ~n_rows / 1024 global atomics instead of one per wave.
constexpr int GR_WG = 1024;
__global__ __launch_bounds__(GR_WG) void k_group_masks(GroupArgs a) {
  __shared__ int s_hist[GR_BUCKETS];
  for (int i = threadIdx.x; i < GR_BUCKETS; i += GR_WG) s_hist[i] = 0;
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * GR_WG + threadIdx.x;
  const bool live = row < a.n_rows;
  unsigned m = 0;
  if (live) {
    for (int k0 = 0; k0 < a.K; k0 += 9) {
      int32_t j[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) j[u] = k0 + u < a.K ? a.nbr[(int64_t)(k0 + u) * a.n_rows + row] : -1;
#pragma unroll
      for (int u = 0; u < 9; ++u) m |= j[u] >= 0 ? (1u << (k0 + u)) : 0u;
    }
    a.masks[row] = (int32_t)m;
  }
  int rank, size, first_lane;
  const int key = group_key(m);
  wave_key_groups(key, live, rank, size, first_lane);
  if (live && rank == 0) atomicAdd(&s_hist[key], size);
  __syncthreads();
  for (int i = threadIdx.x; i < GR_BUCKETS; i += GR_WG)
    if (s_hist[i]) atomicAdd(&a.hist[i], s_hist[i]);
}

__global__ __launch_bounds__(1024) void k_group_scan(GroupArgs a) {
  __shared__ int s_part[1024];
  const int tid = threadIdx.x;
  int v[4], sum = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = a.hist[tid * 4 + u], sum += v[u];
  s_part[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {          // Hillis-Steele inclusive scan of the 1024 partial sums
    const int t = tid >= d ? s_part[tid - d] : 0;
    __syncthreads();
    s_part[tid] += t;
    __syncthreads();
  }
  int run = s_part[tid] - sum;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    a.hist[GR_BUCKETS + tid * 4 + u] = run;
    a.hist[2 * GR_BUCKETS + tid * 4 + u] = 0;
    a.hist[tid * 4 + u] = 0;
    run += v[u];
  }
}

__global__ __launch_bounds__(GR_WG) void k_group_place(GroupArgs a) {
  __shared__ int s_cnt[GR_BUCKETS];          // rows of this workgroup per class, then the workgroup's first position in the class
  for (int i = threadIdx.x; i < GR_BUCKETS; i += GR_WG) s_cnt[i] = 0;
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * GR_WG + threadIdx.x;
  const bool live = row < a.n_rows;
  const int key = live ? group_key((unsigned)a.masks[row]) : 0;
  int rank, size, first_lane;
  wave_key_groups(key, live, rank, size, first_lane);
  int wave_off = 0;
  if (live && rank == 0) wave_off = atomicAdd(&s_cnt[key], size);      // this wave's offset inside the workgroup's share of the class
  wave_off = __shfl(wave_off, first_lane);
  __syncthreads();
  for (int i = threadIdx.x; i < GR_BUCKETS; i += GR_WG) {
    const int c = s_cnt[i];
    if (c) s_cnt[i] = a.hist[GR_BUCKETS + i] + atomicAdd(&a.hist[2 * GR_BUCKETS + i], c);
  }
  __syncthreads();
  const int64_t pos = (int64_t)s_cnt[key] + wave_off + rank;
  if (live && pos >= 0 && pos < a.n_rows) a.perm[pos] = (int32_t)row;      // the range check only matters if the persistent counters were clobbered
}

// cost of the regrouped tiles from the masks alone: active offsets of tile t = popcount(OR of its 16 rows' masks)
__global__ __launch_bounds__(256) void k_tile_cost_masks(const int32_t* __restrict__ masks, const int32_t* __restrict__ perm, int64_t n_rows,
                                                         int64_t n_tiles, uint8_t* __restrict__ cost) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned m = p < n_rows ? (unsigned)masks[perm[p]] : 0u;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) m |= __shfl_xor(m, off);
  const int64_t t = p >> 4;
  if ((threadIdx.x & 15) == 0 && t < n_tiles) {
    const int c = __popc(m);
    cost[t] = (uint8_t)(c > 31 ? 31 : c);
  }
}

extern "C" size_t sv_conv_group_persistent_bytes(void) { return (size_t)3 * GR_BUCKETS * sizeof(int32_t); }

extern "C" int sv_conv_group_rows(const int32_t* nbr, int64_t n_rows, int K, void* persistent, int32_t* masks, int32_t* row_perm,
                                  void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && K > 0 && K <= 27, "sv_conv_group_rows: 1 <= K <= 27 (got %d)", K);
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(nbr && persistent && masks && row_perm, "sv_conv_group_rows: null pointer");
  GroupArgs a;
  a.nbr = nbr, a.n_rows = n_rows, a.K = K, a.masks = masks, a.hist = static_cast<int32_t*>(persistent), a.perm = row_perm;
  const int wgs = sv_div_up(n_rows, GR_WG);
  hipStream_t st = sv_stream(stream);
  hipLaunchKernelGGL(k_group_masks, dim3(wgs), dim3(GR_WG), 0, st, a);
  hipLaunchKernelGGL(k_group_scan, dim3(1), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(k_group_place, dim3(wgs), dim3(GR_WG), 0, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// Work-balanced tile assignment.  A 16-row tile costs as many MFMA steps as it has kernel offsets with at least one neighbour
// (9 / 18 / 27 for the bench layers, depending on how many z-slices it spans); with tiles dealt to waves by position the
// busiest wave had 1.8x the mean work and set the kernel time.  Tiles are counting-sorted by cost and dealt to the waves in
// snake order (max / mean 1.05).  Order inside a cost bucket is arbitrary: every output row is still produced by one wave with
// the same summation order, so results do not depend on it.
// The order is a property of the rulebook table: sv_conv_tile_order computes it once (two small launches), the conv launches
// that use the table pass it in.
struct TileOrderArgs {
  const int32_t* nbr;
  int64_t n_rows, n_tiles, n_waves;
  int K, G;             // G = tiles per wave of the conv kernel that will use the order
  uint8_t* cost;        // scratch: (n_tiles)
  int32_t* hist;        // scratch: [0][32] tiles per cost, [1][32] running fill per cost
  int32_t* tile_of;     // out: [wave * G + slot] -> tile or -1
};

constexpr int TO_WGS = 512;

__global__ __launch_bounds__(256) void k_tile_cost(TileOrderArgs a) {
  __shared__ int s_hist[32];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 32) s_hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t n_groups = (a.n_tiles + 3) / 4;                      // one wave per 4 tiles (64 rows)
  for (int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6; w < n_groups; w += (int64_t)gridDim.x * 4) {
    const int64_t row = w * 64 + lane;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int k0 = 0; k0 < a.K; k0 += 9) {               // 9 independent loads in flight per lane (27 = 3 x 9)
      int32_t j[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) j[u] = (row < a.n_rows && k0 + u < a.K) ? a.nbr[(int64_t)(k0 + u) * a.n_rows + row] : -1;
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const unsigned long long v = __ballot(j[u] >= 0);
        c0 += (v & 0xffffull) != 0, c1 += ((v >> 16) & 0xffffull) != 0, c2 += ((v >> 32) & 0xffffull) != 0, c3 += (v >> 48) != 0;
      }
    }
    if (lane < 4) {
      const int64_t t = w * 4 + lane;
      int c = lane == 0 ? c0 : (lane == 1 ? c1 : (lane == 2 ? c2 : c3));
      c = c > 31 ? 31 : c;
      if (t < a.n_tiles) {
        a.cost[t] = (uint8_t)c;
        atomicAdd(&s_hist[c], 1);
      }
    }
  }
  __syncthreads();
  if (a.hist && threadIdx.x < 32 && s_hist[threadIdx.x]) atomicAdd(&a.hist[threadIdx.x], s_hist[threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_tile_deal(TileOrderArgs a) {
  __shared__ int s_start[32], s_cnt[32], s_base[32];
  if (threadIdx.x < 32) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) {                 // descending cost: the most expensive bucket first
    int acc = 0;
    for (int c = 31; c >= 0; --c) s_start[c] = acc, acc += a.hist[c];
  }
  __syncthreads();
  // this workgroup's contiguous slice of tiles: count per bucket, reserve one range per bucket, then place
  const int64_t per = (a.n_tiles + gridDim.x - 1) / gridDim.x, t0 = blockIdx.x * per, t1 = min(a.n_tiles, t0 + per);
  if (blockIdx.x == 0 && threadIdx.x == 0) a.tile_of[0] = a.G;
  for (int64_t t = t0 + threadIdx.x; t < t1; t += 256) atomicAdd(&s_cnt[a.cost[t]], 1);
  __syncthreads();
  if (threadIdx.x < 32) {
    s_base[threadIdx.x] = s_cnt[threadIdx.x] ? s_start[threadIdx.x] + atomicAdd(&a.hist[32 + threadIdx.x], s_cnt[threadIdx.x]) : 0;
    s_cnt[threadIdx.x] = 0;
  }
  __syncthreads();
  for (int64_t t = t0 + threadIdx.x; t < t1; t += 256) {
    const int c = a.cost[t];
    const int64_t p = s_base[c] + atomicAdd(&s_cnt[c], 1);
    const int64_t r = p / a.n_waves, i = p - r * a.n_waves;
    const int64_t wv = (r & 1) ? a.n_waves - 1 - i : i;          // snake: dense and sparse tiles alternate per wave
    a.tile_of[4 + wv * a.G + r] = (int32_t)t;      // 4-int header: [0] = G
  }
}

// Same result class as k_tile_deal for up to TO_ONE_WG_TILES tiles, in ONE workgroup and without the two fills: the histogram lives
// in LDS, every slot of the order (header, tiles, -1 padding) is written here.
constexpr int TO_ONE_WG_TILES = 65536;
__global__ __launch_bounds__(1024) void k_tile_deal_one(TileOrderArgs a) {
  __shared__ int s_cnt[32], s_start[32];
  const int tid = threadIdx.x;
  if (tid < 32) s_cnt[tid] = 0;
  __syncthreads();
  for (int64_t t = tid; t < a.n_tiles; t += 1024) atomicAdd(&s_cnt[a.cost[t]], 1);
  __syncthreads();
  if (tid == 0) {                          // descending cost: the most expensive bucket first
    int acc = 0;
    for (int c = 31; c >= 0; --c) s_start[c] = acc, acc += s_cnt[c];
  }
  __syncthreads();
  if (tid < 32) s_cnt[tid] = 0;
  if (tid < 4) a.tile_of[tid] = tid == 0 ? a.G : -1;
  __syncthreads();
  const int64_t slots = a.n_waves * a.G;
  for (int64_t t = tid; t < slots; t += 1024) {
    if (t < a.n_tiles) {
      const int c = a.cost[t];
      const int64_t p = s_start[c] + atomicAdd(&s_cnt[c], 1);
      const int64_t r = p / a.n_waves, i = p - r * a.n_waves;
      const int64_t wv = (r & 1) ? a.n_waves - 1 - i : i;          // snake: dense and sparse tiles alternate per wave
      a.tile_of[4 + wv * a.G + r] = (int32_t)t;
    } else {                                                        // positions n_tiles .. slots-1 of the snake stay empty
      const int64_t r = t / a.n_waves, i = t - r * a.n_waves;
      const int64_t wv = (r & 1) ? a.n_waves - 1 - i : i;
      a.tile_of[4 + wv * a.G + r] = -1;
    }
  }
}

// Tiles per wave.  A launch lasts as long as its fullest CU: 526 workgroups (4 waves each) on 256 CUs put 3 on 14 CUs and 2 on
// the rest (1.46x the mean), and a launch with fewer workgroups than CUs leaves one wave per SIMD with nothing to hide its
// operand latency behind.  Model: time ~ ceil(WGs / 256) * G / eff(waves per SIMD), eff = 0.45 / 0.75 / 0.9 for 1 / 2 / >= 3
// resident workgroups per CU (measured shape of the curve on the bench layers), + 4 % per step of G below 4 for the extra
// weight-slab loads.  Measured on the 64->64 layers: 8412 tiles G = 4 230 us, G = 3 183 us; 4012 tiles G = 4 130 us, G = 2 105 us.
// (The same whole-rounds idea applied to the weight-gradient grid made it slower: its workgroups are unequal, dispatch order
// already balances them.)
static int conv_tiles_per_wave(int64_t n_rows) {
  const int64_t n_tiles = (n_rows + 15) / 16;
  int best = 4;
  double best_score = 1e30;
  for (int g = 4; g >= 2; --g) {
    const int64_t wgs = ((n_tiles + g - 1) / g + 3) / 4;
    if (wgs <= 0) continue;
    const int64_t per_cu = (wgs + 255) / 256;
    const double eff = per_cu >= 3 ? 0.9 : (per_cu == 2 ? 0.75 : 0.45);
    const double score = (double)per_cu * g / eff * (1.0 + 0.04 * (4 - g));
    if (score < best_score - 1e-9) best_score = score, best = g;
  }
  return best;
}
// layers with fewer than 32x32 channel products per tile are bound by their operand loads, not by matrix-core time: keep G = 4
static int conv_tiles_per_wave(int64_t n_rows, int Kd, int Nc) { return (Kd / 16) * (Nc / 16) < 4 ? 4 : conv_tiles_per_wave(n_rows); }
extern "C" int sv_conv_tiles_per_wave(int64_t n_rows, int Kd, int Nc) { return conv_tiles_per_wave(n_rows < 0 ? 0 : n_rows, Kd, Nc); }

static size_t tile_order_scratch_bytes(int64_t n_rows) { return (size_t)((n_rows + 15) / 16) + 256 + 64 * sizeof(int32_t); }

extern "C" size_t sv_conv_tile_order_scratch_bytes(int64_t n_rows) { return tile_order_scratch_bytes(n_rows < 0 ? 0 : n_rows); }
extern "C" size_t sv_conv_tile_order_bytes(int64_t n_rows) {
  const int64_t n_tiles = ((n_rows < 0 ? 0 : n_rows) + 15) / 16;
  return (size_t)(((n_tiles + 1) / 2) * 2 + 16) * sizeof(int32_t);     // 4-int header + enough slots for any G in {2, 3, 4}
}

extern "C" int sv_conv_tile_order(const int32_t* nbr, int64_t n_rows, int K, int tiles_per_wave, void* scratch, int32_t* tile_order,
                                  void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && K > 0 && tiles_per_wave >= 2 && tiles_per_wave <= 4, "sv_conv_tile_order: bad sizes (tiles_per_wave %d)", tiles_per_wave);
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(nbr && scratch && tile_order, "sv_conv_tile_order: null pointer");
  TileOrderArgs a{};
  a.nbr = nbr, a.n_rows = n_rows, a.K = K;
  a.n_tiles = (n_rows + 15) / 16;
  a.G = tiles_per_wave;
  a.n_waves = (a.n_tiles + a.G - 1) / a.G;
  a.hist = reinterpret_cast<int32_t*>(scratch);
  a.cost = reinterpret_cast<uint8_t*>(scratch) + 64 * sizeof(int32_t);
  a.tile_of = tile_order;
  hipStream_t st = sv_stream(stream);
  const int wgs = (int)((a.n_tiles + 15) / 16 < TO_WGS ? (a.n_tiles + 15) / 16 : TO_WGS);
  if (a.n_tiles <= TO_ONE_WG_TILES) {
    a.hist = nullptr;
    hipLaunchKernelGGL(k_tile_cost, dim3(wgs), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_tile_deal_one, dim3(1), dim3(1024), 0, st, a);
  } else {
    SV_HIP(hipMemsetAsync(a.hist, 0, 64 * sizeof(int32_t), st));
    SV_HIP(hipMemsetAsync(tile_order, 0xFF, (size_t)(4 + a.n_waves * a.G) * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_tile_cost, dim3(wgs), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_tile_deal, dim3(wgs), dim3(256), 0, st, a);
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_conv_tile_order_grouped(const int32_t* masks, const int32_t* row_perm, int64_t n_rows, int tiles_per_wave, void* scratch,
                                          int32_t* tile_order, void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && tiles_per_wave >= 2 && tiles_per_wave <= 4, "sv_conv_tile_order_grouped: bad sizes");
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(masks && row_perm && scratch && tile_order, "sv_conv_tile_order_grouped: null pointer");
  TileOrderArgs a{};
  a.n_rows = n_rows, a.n_tiles = (n_rows + 15) / 16, a.G = tiles_per_wave;
  SV_CHECK_ARG(a.n_tiles <= TO_ONE_WG_TILES, "sv_conv_tile_order_grouped: at most %d tiles", TO_ONE_WG_TILES);
  a.n_waves = (a.n_tiles + a.G - 1) / a.G;
  a.cost = reinterpret_cast<uint8_t*>(scratch) + 64 * sizeof(int32_t);
  a.tile_of = tile_order;
  hipStream_t st = sv_stream(stream);
  hipLaunchKernelGGL(k_tile_cost_masks, dim3(sv_div_up(a.n_tiles * 16, 256)), dim3(256), 0, st, masks, row_perm, n_rows, a.n_tiles, a.cost);
  hipLaunchKernelGGL(k_tile_deal_one, dim3(1), dim3(1024), 0, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

template <int NT, int KQ, int RS_G>
__global__ __launch_bounds__(256, 3) void k_spconv_rs3(ConvArgs a, uint32_t x_bytes, uint32_t w_bytes) {
  constexpr int Kd = KQ * 16, Nc = NT * 16;
  __shared__ int32_t s_idx_all[4][RS3_KMAX][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const int64_t n_tiles = (a.n_rows + 15) / 16;
  const int64_t n_waves = (n_tiles + RS_G - 1) / RS_G;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wid;
  if (wave_id >= n_waves) return;
  int32_t(*s_idx)[64] = s_idx_all[wid];
  // rows past the end for an empty slot
  auto tile_row0 = [&](int g) {
    if (!a.tile_order || a.tile_order[0] != RS_G) return (wave_id + (int64_t)g * n_waves) * 16;   // by position (strided)
    const int32_t t = a.tile_order[4 + wave_id * RS_G + g];
    return (t >= 0 && t < n_tiles) ? (int64_t)t * 16 : a.n_rows;
  };

  // neighbour indices of the wave's rows -> LDS (lane = (tile lane>>4, row lane&15)); per-offset tile masks in lane k of maskreg
  unsigned maskreg = 0;
  {
    const int g = (lane >> 4) % RS_G;
    const int64_t p = tile_row0(g) + li;
    const bool valid = (lane >> 4) < RS_G && p < a.n_rows;
    const int64_t r = (valid && a.row_perm) ? (int64_t)a.row_perm[p] : p;     // regrouped tiles: 16 arbitrary rows
    for (int k0 = 0; k0 < a.K; k0 += 9) {                  // 9 table reads in flight: 3 load latencies before the first MFMA, not 27
      int32_t jv[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const int k = k0 + u;
        jv[u] = (valid && k < a.K) ? a.nbr[(int64_t)(a.k_flip ? a.K - 1 - k : k) * a.n_rows + r] : -1;
      }
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const int k = k0 + u;
        if (k < a.K) {
          s_idx[k][lane] = jv[u];
          const unsigned long long vote = __ballot(jv[u] >= 0);
          unsigned m = 0;
#pragma unroll
          for (int t = 0; t < RS_G; ++t) m |= ((vote >> (16 * t)) & 0xffffull) ? (1u << t) : 0u;
          if (lane == k) maskreg = m;
        }
      }
    }
  }
  const unsigned long long active = __ballot(maskreg != 0);

  f32x4 acc[RS_G][NT];
#pragma unroll
  for (int g = 0; g < RS_G; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (active) {
    const i32x4 srd_x = make_srd(a.X, x_bytes), srd_w = make_srd(g_wfrag, w_bytes);
    f32x4 A[3][RS_G], B[3][NT];
    // load iterator (two steps ahead of the compute iterator)
    unsigned long long la = active;
    int kl = __ffsll((long long)la) - 1, ql = 0;
    int32_t jl[RS_G];
    auto read_j = [&]() {
#pragma unroll
      for (int g = 0; g < RS_G; ++g) jl[g] = s_idx[kl][g * 16 + li];
    };
    read_j();
    auto issue = [&](f32x4 (&As)[RS_G], f32x4 (&Bs)[NT]) {       // exactly NLOAD loads, always
      const bool live = kl >= 0;
#pragma unroll
      for (int g = 0; g < RS_G; ++g) {
        const uint32_t off = (live && jl[g] >= 0) ? (uint32_t)((jl[g] * Kd + ql * 16 + kk * 4) * 4) : 0xfffffff0u;
        As[g] = buf_load_b128(srd_x, off);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const uint32_t off = live ? (uint32_t)(((((kl * KQ + ql) * NT + t) * 64 + lane) * 4) * 4) : 0xfffffff0u;
        Bs[t] = buf_load_b128(srd_w, off);
      }
      if (live && ++ql == KQ) {
        ql = 0;
        la &= la - 1;
        kl = la ? __ffsll((long long)la) - 1 : -1;
        if (kl >= 0) read_j();
      }
    };
    // compute iterator
    unsigned long long ca = active;
    int kc = __ffsll((long long)ca) - 1, qc = 0;
    unsigned mc = (unsigned)__builtin_amdgcn_readlane((int)maskreg, kc);
    auto compute = [&](f32x4 (&As)[RS_G], f32x4 (&Bs)[NT]) {
      // the stage's loads are the oldest NLOAD in flight: everything issued later (2 steps) may stay outstanding
#define RS3_WAIT(N, ...) asm volatile("s_waitcnt vmcnt(" #N ")" : __VA_ARGS__)
#define V(x) "+v"(x)
      if constexpr (RS_G == 4 && NT == 4) RS3_WAIT(16, V(As[0]), V(As[1]), V(As[2]), V(As[3]), V(Bs[0]), V(Bs[1]), V(Bs[2]), V(Bs[3]));
      else if constexpr (RS_G == 4 && NT == 2) RS3_WAIT(12, V(As[0]), V(As[1]), V(As[2]), V(As[3]), V(Bs[0]), V(Bs[1]));
      else if constexpr (RS_G == 4 && NT == 1) RS3_WAIT(10, V(As[0]), V(As[1]), V(As[2]), V(As[3]), V(Bs[0]));
      else if constexpr (RS_G == 3 && NT == 4) RS3_WAIT(14, V(As[0]), V(As[1]), V(As[2]), V(Bs[0]), V(Bs[1]), V(Bs[2]), V(Bs[3]));
      else if constexpr (RS_G == 3 && NT == 2) RS3_WAIT(10, V(As[0]), V(As[1]), V(As[2]), V(Bs[0]), V(Bs[1]));
      else if constexpr (RS_G == 3 && NT == 1) RS3_WAIT(8, V(As[0]), V(As[1]), V(As[2]), V(Bs[0]));
      else if constexpr (RS_G == 2 && NT == 4) RS3_WAIT(12, V(As[0]), V(As[1]), V(Bs[0]), V(Bs[1]), V(Bs[2]), V(Bs[3]));
      else if constexpr (RS_G == 2 && NT == 2) RS3_WAIT(8, V(As[0]), V(As[1]), V(Bs[0]), V(Bs[1]));
      else RS3_WAIT(6, V(As[0]), V(As[1]), V(Bs[0]));
#undef V
#undef RS3_WAIT
      // per tile: 4 passes over the NT column tiles, so that consecutive MFMAs never share an accumulator (a dependent
      // v_mfma_f32_16x16x4_f32 issues 47 cycles after its producer, an independent one after 32)
#pragma unroll
      for (int g = 0; g < RS_G; ++g)
        if ((mc >> g) & 1u) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(As[g].x, Bs[t].x, acc[g][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(As[g].y, Bs[t].y, acc[g][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(As[g].z, Bs[t].z, acc[g][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(As[g].w, Bs[t].w, acc[g][t], 0, 0, 0);
        }
      if (++qc == KQ) {
        qc = 0;
        ca &= ca - 1;
        kc = ca ? __ffsll((long long)ca) - 1 : -1;
        if (kc >= 0) mc = (unsigned)__builtin_amdgcn_readlane((int)maskreg, kc);
      }
    };
    issue(A[0], B[0]);
    issue(A[1], B[1]);
    while (true) {
      issue(A[2], B[2]);
      compute(A[0], B[0]);
      if (kc < 0) break;
      issue(A[0], B[0]);
      compute(A[1], B[1]);
      if (kc < 0) break;
      issue(A[1], B[1]);
      compute(A[2], B[2]);
      if (kc < 0) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // retire the dummy loads before the registers are reused
  }
  // D layout (16x16): col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int g = 0; g < RS_G; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t p = tile_row0(g) + kk * 4 + r;
        const int col = t * 16 + li;
        if (p < a.n_rows) {
          const int64_t row = a.row_perm ? (int64_t)a.row_perm[p] : p;
          a.Y[row * Nc + col] = conv_epilogue(acc[g][t][r], col, row, a);
        }
      }
}

// shapes the rs3 kernel is instantiated for (the only kernel that takes row_perm / table_k_reversed)
static bool rs3_applies(int K, int Kd, int Nc) {
  if (K > RS3_KMAX || Kd % 16 || Nc % 16 || Kd > 64 || Nc > 64 || Kd < 16 || Nc < 16) return false;
  const int nt = Nc / 16, kq = Kd / 16;
  return nt != 3 && kq != 3 && !(nt == 4 && kq == 3) && !(nt == 3 && kq == 4);
}
extern "C" int sv_conv_mfma_kernel_applies(int K, int Kd, int Nc) { return rs3_applies(K, Kd, Nc) ? 1 : 0; }

static int try_launch_rs3(const ConvArgs& a, const WStride& ws, int64_t n_src, hipStream_t st) {
  if (!rs3_applies(a.K, a.Kd, a.Nc)) return -1;
  const uint64_t xb = (uint64_t)n_src * a.Kd * 4, wb = (uint64_t)a.K * a.Nc * a.Kd * 4;
  if (xb >= 0xfffffff0ull || wb >= 0xfffffff0ull) return -1;
  const int64_t n_tiles = (a.n_rows + 15) / 16;
  const int G = conv_tiles_per_wave(a.n_rows, a.Kd, a.Nc);   // the value the caller passed to sv_conv_tile_order for a.tile_order
  const int64_t n_waves = (n_tiles + G - 1) / G;
  const dim3 grid((unsigned)((n_waves + 3) / 4));
  const int nt = a.Nc / 16, kq = a.Kd / 16;
  float* wf = nullptr;
  if (hipGetSymbolAddress(reinterpret_cast<void**>(&wf), HIP_SYMBOL(g_wfrag)) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_weight_fragments, dim3(sv_grid_1d((int64_t)a.K * a.Nc * a.Kd / 4, 256)), dim3(256), 0, st, a.Wt, ws, a.K, a.Nc, a.Kd, wf);
#define RS3_CASE(NTV, KQV)                                                                                                       \
  if (nt == NTV && kq == KQV) {                                                                                                  \
    if (G == 4) hipLaunchKernelGGL((k_spconv_rs3<NTV, KQV, 4>), grid, dim3(256), 0, st, a, (uint32_t)xb, (uint32_t)wb);          \
    else if (G == 3) hipLaunchKernelGGL((k_spconv_rs3<NTV, KQV, 3>), grid, dim3(256), 0, st, a, (uint32_t)xb, (uint32_t)wb);     \
    else hipLaunchKernelGGL((k_spconv_rs3<NTV, KQV, 2>), grid, dim3(256), 0, st, a, (uint32_t)xb, (uint32_t)wb);                 \
    return 0;                                                                                                                    \
  }
  RS3_CASE(4, 4) RS3_CASE(4, 2) RS3_CASE(2, 4) RS3_CASE(2, 2) RS3_CASE(2, 1) RS3_CASE(1, 2) RS3_CASE(1, 1) RS3_CASE(4, 1) RS3_CASE(1, 4)
#undef RS3_CASE
  return -1;
}

template <int NT, int G>
static int launch_rs_kq(const ConvArgs& a, int kq, hipStream_t st) {
  const int64_t n_tiles = (a.n_rows + 15) / 16;
  const int64_t n_waves = (n_tiles + G - 1) / G;
  const dim3 grid((unsigned)((n_waves + 3) / 4));
  switch (kq) {
    case 1: hipLaunchKernelGGL((k_spconv_rs<NT, 1, G>), grid, dim3(256), 0, st, a); return 0;
    case 2: hipLaunchKernelGGL((k_spconv_rs<NT, 2, G>), grid, dim3(256), 0, st, a); return 0;
    case 4: hipLaunchKernelGGL((k_spconv_rs<NT, 4, G>), grid, dim3(256), 0, st, a); return 0;
    case 8: hipLaunchKernelGGL((k_spconv_rs<NT, 8, G>), grid, dim3(256), 0, st, a); return 0;
  }
  return -1;
}

static int try_launch_rs(const ConvArgs& a, hipStream_t st) {
  if (a.Kd % 16 || a.Nc % 16 || a.K > 64) return -1;
  switch (a.Nc / 16) {
    case 1: return launch_rs_kq<1, 4>(a, a.Kd / 16, st);
    case 2: return launch_rs_kq<2, 4>(a, a.Kd / 16, st);
    case 4: return launch_rs_kq<4, 4>(a, a.Kd / 16, st);
    case 8: return launch_rs_kq<8, 4>(a, a.Kd / 16, st);
  }
  return -1;
}

// Generic VALU path for channel counts the MFMA tiling does not cover (e.g. the C_in = 3 input layer):
// one thread per (row, 4 output columns), weights read through L1/L2.
__global__ __launch_bounds__(256) void k_spconv_valu(ConvArgs a) {
  const int nq = (a.Nc + 3) / 4;
  const int64_t total = a.n_rows * nq;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / nq;
    const int n0 = (int)(idx - row * nq) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < a.K; ++k) {
      const int32_t j = a.nbr[(int64_t)k * a.n_rows + row];
      if (j < 0) continue;
      const float* x = a.X + (int64_t)j * a.Kd;
      for (int u = 0; u < 4; ++u) {
        if (n0 + u >= a.Nc) break;
        const float* w = a.Wt + ((int64_t)k * a.Nc + n0 + u) * a.Kd;
        float s = acc[u];
        for (int c = 0; c < a.Kd; ++c) s = fmaf(x[c], w[c], s);
        acc[u] = s;
      }
    }
    for (int u = 0; u < 4 && n0 + u < a.Nc; ++u) a.Y[row * a.Nc + n0 + u] = conv_epilogue(acc[u], n0 + u, row, a);
  }
}

static int gather_gemm_impl(const float* X, int64_t n_src, const int32_t* nbr, const float* Wt, const int64_t* w_strides, float* Y, int64_t n_rows,
                            int K, int Kd, int Nc, const float* bias, const float* scale, const float* shift, const float* residual, int relu,
                            const int32_t* tile_order, const int32_t* row_perm, int table_k_reversed, void* stream);

extern "C" int sv_sparse_conv_gather_gemm(const float* X, int64_t n_src, const int32_t* nbr, const float* Wt, float* Y,
                                          int64_t n_rows, int K, int Kd, int Nc, const float* bias, const float* scale, const float* shift,
                                          const float* residual, int relu, void* stream) {
  return gather_gemm_impl(X, n_src, nbr, Wt, nullptr, Y, n_rows, K, Kd, Nc, bias, scale, shift, residual, relu, nullptr, nullptr, 0, stream);
}

extern "C" int sv_sparse_conv_gather_gemm_ordered(const float* X, int64_t n_src, const int32_t* nbr, const float* Wt, float* Y,
                                                  int64_t n_rows, int K, int Kd, int Nc, const float* bias, const float* scale,
                                                  const float* shift, const float* residual, int relu, const int32_t* tile_order,
                                                  void* stream) {
  return gather_gemm_impl(X, n_src, nbr, Wt, nullptr, Y, n_rows, K, Kd, Nc, bias, scale, shift, residual, relu, tile_order, nullptr, 0, stream);
}

extern "C" int sv_sparse_conv_gather_gemm_strided(const float* X, int64_t n_src, const int32_t* nbr, const float* W, int64_t w_stride_k,
                                                  int64_t w_stride_n, int64_t w_stride_c, float* Y, int64_t n_rows, int K, int Kd, int Nc,
                                                  const float* bias, const float* scale, const float* shift, const float* residual, int relu,
                                                  const int32_t* tile_order, const int32_t* row_perm, int table_k_reversed, void* stream) {
  const int64_t ws[3] = {w_stride_k, w_stride_n, w_stride_c};
  return gather_gemm_impl(X, n_src, nbr, W, ws, Y, n_rows, K, Kd, Nc, bias, scale, shift, residual, relu, tile_order, row_perm,
                          table_k_reversed, stream);
}

// Input layer (C_in = 3 or 4 point features -> 16 channels, spconv_backbone.py:77-81): HBM-bound -- 4*K bytes of neighbour table and
// 4*Nc bytes of output per row against 2*K*Kd*Nc flops.  Same thread mapping and summation order as k_spconv_valu (bit-identical
// results); the weights (K*Nc*Kd floats, 5 KB) are staged in LDS and the K table reads of a row are issued 9 at a time.
constexpr int SC_SMALL_KD = 4, SC_SMALL_LDS = 27 * 32 * SC_SMALL_KD;
template <int KD>
__global__ __launch_bounds__(256) void k_spconv_small_cin(ConvArgs a) {
  __shared__ float s_w[SC_SMALL_LDS];
  for (int i = threadIdx.x; i < a.K * a.Nc * KD; i += 256) s_w[i] = a.Wt[i];
  __syncthreads();
  const int nq = (a.Nc + 3) / 4;
  const int64_t total = a.n_rows * nq;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / nq;
    const int n0 = (int)(idx - row * nq) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < a.K; k0 += 9) {
      int32_t j[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) j[u] = k0 + u < a.K ? a.nbr[(int64_t)(k0 + u) * a.n_rows + row] : -1;
      float x[9][KD];
#pragma unroll
      for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int c = 0; c < KD; ++c) x[u][c] = j[u] >= 0 ? a.X[(int64_t)j[u] * KD + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        if (j[u] < 0) continue;
        const float* w = s_w + (size_t)((k0 + u) * a.Nc + n0) * KD;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (n0 + v >= a.Nc) break;
          float t = acc[v];
#pragma unroll
          for (int c = 0; c < KD; ++c) t = fmaf(x[u][c], w[v * KD + c], t);
          acc[v] = t;
        }
      }
    }
    for (int v = 0; v < 4 && n0 + v < a.Nc; ++v) a.Y[row * a.Nc + n0 + v] = conv_epilogue(acc[v], n0 + v, row, a);
  }
}

static int gather_gemm_impl(const float* X, int64_t n_src, const int32_t* nbr, const float* Wt, const int64_t* w_strides, float* Y, int64_t n_rows,
                            int K, int Kd, int Nc, const float* bias, const float* scale, const float* shift, const float* residual, int relu,
                            const int32_t* tile_order, const int32_t* row_perm, int table_k_reversed, void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && K > 0 && Kd > 0 && Nc > 0, "sparse_conv: bad sizes");
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(X && nbr && Wt && Y, "sparse_conv: null pointer");
  SV_CHECK_ARG((scale == nullptr) == (shift == nullptr), "sparse_conv: scale and shift go together");
  ConvArgs a{X, nbr, Wt, Y, bias, scale, shift, residual, relu, n_rows, K, Kd, Nc, tile_order, row_perm, table_k_reversed};
  hipStream_t st = sv_stream(stream);
  const int nt = Nc / 16;
  WStride ws{(int64_t)Nc * Kd, (int64_t)Kd, 1};
  if (w_strides) ws = WStride{w_strides[0], w_strides[1], w_strides[2]};
  const bool w_packed = ws.k == (int64_t)Nc * Kd && ws.n == Kd && ws.c == 1;
  static const bool force_v1 = getenv("SEEVCN_SPCONV_V1") != nullptr;
  static const bool use_rs3 = getenv("SEEVCN_SPCONV_NORS3") == nullptr;
  // the rs3 path re-lays the weights into fragment order anyway: it reads any (K, Nc, Kd) view through its strides
  if ((Kd % 16 == 0) && (Nc % 16 == 0) && ((uintptr_t)X % 16 == 0) && !force_v1 && use_rs3 && try_launch_rs3(a, ws, n_src, st) == 0) {
    SV_LAUNCH_CHECK();
    return SV_OK;
  }
  SV_CHECK_ARG(!row_perm && !table_k_reversed, "sparse_conv: row_perm / table_k_reversed (grouped table) need the rs3 kernel: C_in, C_out multiples of 16 up to 64, K <= %d", RS3_KMAX);
  if (!w_packed) {                          // the other kernels read a contiguous (K, Nc, Kd) array
    const int64_t total = (int64_t)K * Nc * Kd;
    SV_CHECK_ARG(total <= WPACK_FLOATS, "sparse_conv: a strided weight view of %lld floats does not fit the pack buffer; pass it contiguous", (long long)total);
    float* wp = nullptr;
    SV_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&wp), HIP_SYMBOL(g_wpack)));
    hipLaunchKernelGGL(k_weight_pack, dim3(sv_grid_1d(total, 256)), dim3(256), 0, st, Wt, ws, K, Nc, Kd, wp);
    a.Wt = Wt = wp;
  }
  const bool mfma_ok = (Kd % 16 == 0) && (Nc % 16 == 0) && (nt == 1 || nt == 2 || nt == 4 || nt == 8) &&
                       ((uintptr_t)X % 16 == 0) && ((uintptr_t)Wt % 16 == 0);
  if (mfma_ok && !force_v1 && try_launch_rs(a, st) == 0) {
    SV_LAUNCH_CHECK();
    return SV_OK;
  }
  if (mfma_ok) {
    const int grid = sv_div_up(n_rows, SC_ROWS_PER_BLOCK);
    const size_t lds = (size_t)Nc * ((Kd < SC_KSLICE ? Kd : SC_KSLICE) + 4) * sizeof(float);
    switch (nt) {
      case 1: hipLaunchKernelGGL(k_spconv_mfma<1>, dim3(grid), dim3(SC_THREADS), lds, st, a); break;
      case 2: hipLaunchKernelGGL(k_spconv_mfma<2>, dim3(grid), dim3(SC_THREADS), lds, st, a); break;
      case 4: hipLaunchKernelGGL(k_spconv_mfma<4>, dim3(grid), dim3(SC_THREADS), lds, st, a); break;
      default: hipLaunchKernelGGL(k_spconv_mfma<8>, dim3(grid), dim3(SC_THREADS), lds, st, a); break;
    }
  } else if ((Kd == 3 || Kd == 4) && K * Nc * Kd <= SC_SMALL_LDS) {
    const dim3 grid(sv_grid_1d(n_rows * ((Nc + 3) / 4), 256, 256 * 8));
    if (Kd == 3) hipLaunchKernelGGL(k_spconv_small_cin<3>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_spconv_small_cin<4>, grid, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(k_spconv_valu, dim3(sv_grid_1d(n_rows * ((Nc + 3) / 4), 256, 256 * 16)), dim3(256), 0, st, a);
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: dW[k][c][n] = sum_o X[nbr[k][o]][c] * dY[o][n]   (reduction over rows)
// Stage 1: each workgroup reduces a chunk of rows for one offset k into a partial (Cin x Cout) slab
//          (MFMA 16x16x4 with the row index as the contraction dimension); stage 2 sums the slabs in a
//          fixed order -> bitwise reproducible, no atomics.
// ------------------------------------------------------------------------------------------------
constexpr int WG_CHUNK_MAX = 4096;  // rows per stage-1 workgroup (upper bound; sized per launch to fill the chip)
constexpr int WG_SUB = 4;           // 64-row blocks a wave compacts per pass
constexpr int WG_DEPTH = 4;         // operand ring depth (MFMA steps)

struct WgradArgs;
static int wgrad_chunk_rows(int64_t n_rows, int K, int groups) {
  // aim at ~2048 workgroups: chunk = n_rows*K*groups/2048 rounded up to one pass of the four waves (4 x WG_SUB x 64 rows)
  constexpr int64_t pass = 256 * WG_SUB;
  int64_t c = (n_rows * K * groups + 2047) / 2048;
  c = (c + pass - 1) / pass * pass;
  if (c < pass) c = pass;
  if (c > WG_CHUNK_MAX) c = WG_CHUNK_MAX;
  return (int)c;
}

struct WgradArgs {
  const float* X;        // (n_src, Cin)
  const int32_t* nbr;    // (K, n_rows)
  const float* dY;       // (n_rows, Cout)
  float* partial;        // (nchunks, K, Cin, Cout)
  int64_t n_rows;
  int K, Cin, Cout, nchunks, chunk_rows;
};

template <int N> struct WgVec;
template <> struct WgVec<4> { using type = f32x4; };
template <> struct WgVec<2> { using type = f32x2; };
template <> struct WgVec<1> { using type = float; };
__device__ __forceinline__ void wg_gload(f32x4& v, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void wg_gload(f32x2& v, const float* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void wg_gload(float& v, const float* p) { asm volatile("global_load_dword %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ float wg_elem(const f32x4& v, int i) { return v[i]; }
__device__ __forceinline__ float wg_elem(const f32x2& v, int i) { return v[i]; }
__device__ __forceinline__ float wg_elem(const float& v, int) { return v; }

// grid = (nchunks, K, tile groups).  Each wave walks its share of the chunk 64 rows at a time: one coalesced read of the
// neighbour table, ballot + prefix popcount compaction of the valid (row, source) pairs into a wave-private LDS list,
// then MFMAs over the COMPACTED pairs only (4 pairs per 16x16x4 step) with the next step's operands requested first.
// The four waves' accumulators are summed through LDS in a fixed order and one slab per (chunk, k) is stored.
template <int CT, int NTL>  // register tile grid: CT x NTL tiles of 16x16 (rows = c_in, cols = c_out)
__global__ __launch_bounds__(256) void k_spconv_wgrad(WgradArgs a) {
  __shared__ int32_t pj[4][64 * WG_SUB], pr[4][64 * WG_SUB];
  __shared__ float red[CT * NTL * 256];
  const int k = blockIdx.y, chunk = blockIdx.x;
  const int ngroups_n = (a.Cout / 16) / NTL;
  const int c_base = (blockIdx.z / ngroups_n) * CT * 16, n_base = (blockIdx.z % ngroups_n) * NTL * 16;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, kk = lane >> 4;
  f32x4 acc[CT][NTL];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < NTL; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t r_begin = (int64_t)chunk * a.chunk_rows, r_end = min(r_begin + (int64_t)a.chunk_rows, a.n_rows);
  const int32_t* nb = a.nbr + (int64_t)k * a.n_rows;

  // Operand fetch for one MFMA step (4 pairs): lane (li, kk) needs X[j_kk][.] for CT tiles and dY[r_kk][.] for NTL tiles.
  // Tile c holds the channels c_base + CT*i + c (i = 0..15), so the CT values of a lane are CONTIGUOUS: one 4*CT-byte load per
  // operand instead of CT scalar loads, and the 16 lanes of a pair read its whole 64*CT-byte row segment.
  // The loads are inline asm: hipcc sinks a plain prefetch load into the block of its first use (measured: load, vmcnt(0), MFMA),
  // an asm load stays where it is written and is retired by the counted s_waitcnt in `consume`.  Steps past the end of the list
  // re-read its last pair (a cache hit); their operands are zeroed after the wait.
  using XV = typename WgVec<CT>::type;
  using YV = typename WgVec<NTL>::type;
  const bool x_in = CT > 1 || c_base + li < a.Cin;           // the 3-channel input layer runs with zero-padded rows
  auto issue = [&](int p, int cnt, XV& xs, YV& ys) {
    const int pc = p < cnt ? p : cnt - 1;
    const int32_t j = pj[wid][pc];
    const int32_t r = pr[wid][pc];
    wg_gload(xs, a.X + (int64_t)j * a.Cin + (x_in ? c_base + CT * li : 0));
    wg_gload(ys, a.dY + ((int64_t)r_begin + r) * a.Cout + n_base + NTL * li);
  };
  // waits for the two loads of this step (the 3 younger steps stay in flight), then 16 x CT x NTL MFMAs
  auto consume = [&](int p0, int cnt, XV& xs, YV& ys) {
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(xs), "+v"(ys));
    if (p0 >= cnt) return;                                   // wave-uniform: a dummy step of the ring's tail
    const bool ok = p0 + kk < cnt;
    float xa[CT], yb[NTL];
#pragma unroll
    for (int c = 0; c < CT; ++c) xa[c] = (ok && x_in) ? wg_elem(xs, c) : 0.f;
#pragma unroll
    for (int t = 0; t < NTL; ++t) yb[t] = ok ? wg_elem(ys, t) : 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int t = 0; t < NTL; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[c], yb[t], acc[c][t], 0, 0, 0);
  };

  for (int64_t base = r_begin + wid * (64 * WG_SUB); base < r_end; base += 256 * WG_SUB) {
    // WG_SUB x 64 rows per wave and pass: the neighbour reads are in flight together and the start-up latency of a pass
    // (table read -> compaction -> first operand loads) is paid once per ~80 pairs instead of once per ~20
    int32_t jv[WG_SUB];
#pragma unroll
    for (int s = 0; s < WG_SUB; ++s) {
      const int64_t r = base + s * 64 + lane;
      jv[s] = r < r_end ? nb[r] : -1;
    }
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < WG_SUB; ++s) {
      const unsigned long long m = __ballot(jv[s] >= 0);
      if (jv[s] >= 0) {
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        pj[wid][pos] = jv[s];
        pr[wid][pos] = (int32_t)(base + s * 64 + lane - r_begin);
      }
      cnt += __popcll(m);
    }
    if (cnt == 0) continue;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-private list: LDS ops of one wave complete in order
    // operand ring, 4 steps deep: step s uses stage s % 4 while the loads of steps s+1 .. s+3 are in flight (a gathered row
    // takes ~2 us under load, a step's MFMAs 0.2 us).  EVERY ring slot issues exactly two loads and every consume waits for
    // vmcnt(6): no conditional issue, so each stage register has one definition per slot and hipcc never copies a stage whose
    // load is still in flight (a copied stage lets the late load land in a register that has been handed to something else).
    static_assert(WG_DEPTH == 4, "the wait count in consume() is written for a 4-deep ring");
    XV x0, x1, x2, x3;
    YV y0, y1, y2, y3;
    issue(kk, cnt, x0, y0);
    issue(4 + kk, cnt, x1, y1);
    issue(8 + kk, cnt, x2, y2);
    for (int p0 = 0; p0 < cnt; p0 += 16) {
      issue(p0 + 12 + kk, cnt, x3, y3);
      consume(p0, cnt, x0, y0);
      issue(p0 + 16 + kk, cnt, x0, y0);
      consume(p0 + 4, cnt, x1, y1);
      issue(p0 + 20 + kk, cnt, x1, y1);
      consume(p0 + 8, cnt, x2, y2);
      issue(p0 + 24 + kk, cnt, x2, y2);
      consume(p0 + 12, cnt, x3, y3);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2));   // retire the tail's dummy loads
  }
  // fixed-order reduction over the 4 waves (wave 0 stores, waves 1..3 add in turn), then one slab per (chunk, k)
  for (int w = 0; w < 4; ++w) {
    if (wid == w) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = &red[((c * NTL + t) * 4 + r) * 64 + lane];
            *d = (w == 0 ? 0.f : *d) + acc[c][t][r];
          }
    }
    __syncthreads();
  }
  // D layout of tile (c,t): col = lane&15, row = 4*(lane>>4) + reg; tile row i is channel c_base + CT*i + c, tile col i is
  // column n_base + NTL*i + t (the interleaved tiles of load_ops)
  float* out = a.partial + (((int64_t)chunk * a.K + k) * a.Cin) * a.Cout;
  for (int e = tid; e < CT * NTL * 256; e += 256) {
    const int ln = e & 63, r = (e >> 6) & 3, tile = e >> 8;
    const int c = tile / NTL, t = tile - c * NTL;
    const int crow = c_base + CT * ((ln >> 4) * 4 + r) + c;
    if (crow < a.Cin) out[(int64_t)crow * a.Cout + n_base + NTL * (ln & 15) + t] = red[e];
  }
}

// generic (any Cin/Cout) stage 1: one thread per (c, n) element, rows of the chunk streamed
__global__ __launch_bounds__(256) void k_spconv_wgrad_valu(WgradArgs a) {
  const int k = blockIdx.y, chunk = blockIdx.x;
  const int64_t r_begin = (int64_t)chunk * a.chunk_rows, r_end = min(r_begin + (int64_t)a.chunk_rows, a.n_rows);
  const int32_t* nb = a.nbr + (int64_t)k * a.n_rows;
  float* out = a.partial + (((int64_t)chunk * a.K + k) * a.Cin) * a.Cout;  // one slab per chunk on this path
  for (int e = threadIdx.x; e < a.Cin * a.Cout; e += blockDim.x) {
    const int c = e / a.Cout, n = e - c * a.Cout;
    float s = 0.f;
    for (int64_t r = r_begin; r < r_end; ++r) {
      const int32_t j = nb[r];
      if (j >= 0) s = fmaf(a.X[(int64_t)j * a.Cin + c], a.dY[r * a.Cout + n], s);
    }
    out[e] = s;
  }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial, int nchunks, int64_t slab, float* __restrict__ dW) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < slab; e += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < nchunks; ++c) s += partial[(int64_t)c * slab + e];
    dW[e] = s;
  }
}

// slab % 4 == 0: 64 float4 columns x 4 quarters of the chunk range per workgroup -- four times the loads in flight of the scalar
// kernel, partial sums combined in a fixed order ((q0 + q1) + (q2 + q3)): still bitwise reproducible
__global__ __launch_bounds__(256) void k_wgrad_reduce4(const float* __restrict__ partial, int nchunks, int64_t slab4, float* __restrict__ dW) {
  __shared__ f32x4 s_q[4][64];
  const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + col;
  const int c0 = (int)((int64_t)nchunks * q / 4), c1 = (int)((int64_t)nchunks * (q + 1) / 4);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (e < slab4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(partial) + e;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) s += __builtin_nontemporal_load(p + (int64_t)c * slab4);
  }
  s_q[q][col] = s;
  __syncthreads();
  if (q == 0 && e < slab4) reinterpret_cast<f32x4*>(dW)[e] = (s_q[0][col] + s_q[1][col]) + (s_q[2][col] + s_q[3][col]);
}

extern "C" size_t sv_sparse_conv_wgrad_scratch_bytes(int64_t n_rows, int K, int Cin, int Cout) {
  const int64_t nchunks = (n_rows + 255) / 256;   // worst case: smallest chunk
  return (size_t)(nchunks > 0 ? nchunks : 1) * K * Cin * Cout * sizeof(float);
}

template <int CT, int NTL>
static void launch_wgrad(const WgradArgs& a, hipStream_t st) {
  const int groups = (((a.Cin + 15) / 16) / CT) * ((a.Cout / 16) / NTL);
  hipLaunchKernelGGL((k_spconv_wgrad<CT, NTL>), dim3(a.nchunks, a.K, groups), dim3(256), 0, st, a);
}

extern "C" int sv_sparse_conv_wgrad(const float* X, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K,
                                    int Cin, int Cout, void* scratch, void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && K > 0 && Cin > 0 && Cout > 0 && dW, "sparse_conv_wgrad: bad arguments");
  hipStream_t st = sv_stream(stream);
  const int64_t slab = (int64_t)K * Cin * Cout;
  if (n_rows == 0) {
    SV_HIP(hipMemsetAsync(dW, 0, (size_t)slab * 4, st));
    return SV_OK;
  }
  SV_CHECK_ARG(X && nbr && dY && scratch, "sparse_conv_wgrad: null pointer");
  const int ct = (Cin + 15) / 16, nt = Cout / 16;
  int tiles_c = 1, tiles_n = 1;
  // C_in that is not a multiple of 16 (the 3-channel input layer) runs on the MFMA path with zero-padded rows
  const bool mfma = Cout % 16 == 0 && (Cin % 16 == 0 || Cin < 16);
  if (mfma) {
    if (ct % 4 == 0 && nt % 4 == 0) { tiles_c = 4; tiles_n = 4; }
    else if (ct % 2 == 0 && nt % 4 == 0) { tiles_c = 2; tiles_n = 4; }
    else if (ct % 2 == 0 && nt % 2 == 0) { tiles_c = 2; tiles_n = 2; }
    else if (nt % 2 == 0) { tiles_c = 1; tiles_n = 2; }
  }
  const int groups = mfma ? (ct / tiles_c) * (nt / tiles_n) : 1;
  const int chunk_rows = wgrad_chunk_rows(n_rows, K, groups);
  WgradArgs a{X, nbr, dY, reinterpret_cast<float*>(scratch), n_rows, K, Cin, Cout, (int)((n_rows + chunk_rows - 1) / chunk_rows), chunk_rows};
  int nslabs = a.nchunks;
  if (mfma) {
    if (tiles_c == 4) launch_wgrad<4, 4>(a, st);
    else if (tiles_c == 2 && tiles_n == 4) launch_wgrad<2, 4>(a, st);
    else if (tiles_c == 2) launch_wgrad<2, 2>(a, st);
    else if (tiles_n == 2) launch_wgrad<1, 2>(a, st);
    else launch_wgrad<1, 1>(a, st);
  } else {
    hipLaunchKernelGGL(k_spconv_wgrad_valu, dim3(a.nchunks, K), dim3(256), 0, st, a);
    nslabs = a.nchunks;
  }
  if (slab % 4 == 0 && (uintptr_t)dW % 16 == 0 && (uintptr_t)a.partial % 16 == 0)
    hipLaunchKernelGGL(k_wgrad_reduce4, dim3(sv_div_up(slab / 4, 64)), dim3(256), 0, st, a.partial, nslabs, slab / 4, dW);
  else
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(sv_grid_1d(slab, 256)), dim3(256), 0, st, a.partial, nslabs, slab, dW);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
