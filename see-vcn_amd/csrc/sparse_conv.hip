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
// LDS-DMA variant for 64 -> 64 channels (the layers that dominate the step).  Same ownership as k_spconv_rs — a wave owns
// 4 strided 16-row tiles x all 64 columns, no workgroup barrier anywhere — but
//   * offset-major: the 64x64 weight slab of an offset is loaded once per wave into 64 registers and serves all of the wave's
//     tiles that have a neighbour at that offset (k_spconv_rs re-requested a quarter slab per 16-channel step);
//   * the gathered rows arrive by LDS-DMA (global_load_lds_dwordx4: 4 whole 256-byte rows per instruction, the per-lane
//     source address makes it a row gather) into a wave-private 2-slot ring, one position ahead of the matrix core, without
//     occupying registers; the LDS image is XOR-swizzled on the SOURCE side (chunk ^ row) so that the MFMA operand reads
//     (ds_read_b128, lane = (row, k-quarter)) are conflict free; the contraction index is permuted (k = 16s + 4q + j) so one
//     b128 feeds 4 MFMAs of each of the 4 column tiles (64 MFMAs per position, 4 independent accumulation chains);
//   * the wave's neighbour indices are parked in LDS once (27 x 64 ints), so the loop issues no register-destination loads
//     except the weight slab at an offset change.
// An earlier version split the columns over the 4 waves of a workgroup (shared rows, 16 MFMAs per wave and position, one
// s_barrier per position): 304 us for the layer that k_spconv_rs does in 248 us — a 512-cycle position cannot carry a
// barrier, ~23 branches and three LDS round trips.  Lessons kept here: LDS reads of the loop are inline asm (for a
// compiler-visible ds_read of memory an LDS-DMA may be writing hipcc waits vmcnt(0)), accumulators are only ever touched in
// statically indexed code (a switch over 8 accumulators became copies through phi registers at every step).
// Summation order per output element is fixed -> bitwise reproducible.
// ------------------------------------------------------------------------------------------------
constexpr int DW_G = 4;            // 16-row tiles per wave
constexpr int DW_KMAX = 27;
constexpr int DW_RING = 3;         // 4 KB position buffers per wave: DW_RING - 1 gathers in flight ahead of the matrix core

__device__ __attribute__((aligned(256))) float g_zero_row[64];

typedef __attribute__((address_space(1))) const void* sv_gptr_t;
typedef __attribute__((address_space(3))) void* sv_lptr_t;

__device__ __forceinline__ uint32_t dm_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ f32x4 dm_read_b128(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ int dm_read_b32(uint32_t addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
#define DM_WAIT_LGKM4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
// register-destination load hidden from hipcc's waitcnt insertion (beside LDS-DMA it would wait vmcnt(0) at the first use):
// the caller counts the queue by hand and waits with dm_wait_vmcnt_x4 before touching the result
__device__ __forceinline__ f32x4 dm_global_load_b128(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void dm_wait_vmcnt_x4(int n) {   // wave-uniform: at most 4*n loads outstanding, n = 0..6
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
  }
}

__global__ __launch_bounds__(256, 2) void k_spconv_dmaw64(ConvArgs a, int64_t n_waves) {
  // one __shared__ object (75 KB: two workgroups per CU); per wave: ring [DW_RING][16 rows x 64 ch] floats | idx int32 [DW_KMAX][64]
  constexpr int RING_BYTES = DW_RING * 4096;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * (RING_BYTES + DW_KMAX * 64 * 4)];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 15, q = lane >> 4;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wid;
  if (wave_id >= n_waves) return;                      // no barrier is ever used: waves are independent
  unsigned char* my = smem + wid * RING_BYTES;
  float(*s_ring)[1024] = reinterpret_cast<float(*)[1024]>(my);
  int32_t(*s_idx)[64] = reinterpret_cast<int32_t(*)[64]>(smem + 4 * RING_BYTES + wid * (DW_KMAX * 64 * 4));
  auto tile_row0 = [&](int g) { return (wave_id + (int64_t)g * n_waves) * 16; };

  // ---- neighbour indices of the wave's 64 rows -> LDS; 4-bit tile mask per offset kept in lane k of `maskreg`
  unsigned maskreg = 0;
  {
    const int64_t row = tile_row0(q) + li;             // lane <-> (tile q, row li)
    const bool valid = row < a.n_rows;
    int32_t j[DW_KMAX];
#pragma unroll
    for (int k = 0; k < DW_KMAX; ++k) j[k] = (valid && k < a.K) ? a.nbr[(int64_t)k * a.n_rows + row] : -1;
#pragma unroll
    for (int k = 0; k < DW_KMAX; ++k) {
      s_idx[k][lane] = j[k];
      const unsigned long long vote = __ballot(j[k] >= 0);
      unsigned m = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t) m |= ((vote >> (16 * t)) & 0xffffull) ? (1u << t) : 0u;
      if (lane == k) maskreg = m;
    }
  }
  unsigned long long kbits = __ballot(maskreg != 0);   // offsets with at least one neighbour

  f32x4 acc[DW_G][4];
#pragma unroll
  for (int g = 0; g < DW_G; ++g)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (kbits) {
    const uint32_t idx_base = dm_lds_addr(&s_idx[0][0]);
    const uint32_t ring_base = dm_lds_addr(&s_ring[0][li * 64]);
    const uint32_t o0 = (uint32_t)(((0 + q) ^ li) << 4), o1 = (uint32_t)(((4 + q) ^ li) << 4), o2 = (uint32_t)(((8 + q) ^ li) << 4),
                   o3 = (uint32_t)(((12 + q) ^ li) << 4);
    // gather of position (k, g) into ring slot `slot`: 4 instructions x 4 rows; lane -> (row 4i + q, position li)
    auto gather = [&](int k, int g, int slot) {
      const uint32_t ib = idx_base + (uint32_t)((k * 64 + g * 16 + q) << 2);
      int32_t j0 = dm_read_b32(ib), j1 = dm_read_b32(ib + 16), j2 = dm_read_b32(ib + 32), j3 = dm_read_b32(ib + 48);
      DM_WAIT_LGKM4(j0, j1, j2, j3);
      const int32_t jj[4] = {j0, j1, j2, j3};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = i * 4 + q;
        const float* src = (jj[i] >= 0 ? a.X + (int64_t)jj[i] * 64 : g_zero_row) + ((li ^ r) << 2);
        __builtin_amdgcn_global_load_lds((sv_gptr_t)src, (sv_lptr_t)&s_ring[slot][i * 256], 16, 0, 0);
      }
    };
    auto load_b = [&](int k, f32x4 (&B)[4][4]) {    // 16 loads, invisible to hipcc's wait insertion
      const float* w = a.Wt + ((int64_t)k * 64 + li) * 64 + q * 4;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) B[t][s] = dm_global_load_b128(w + (int64_t)t * 16 * 64 + s * 16);
    };

    // gather iterator: runs DW_RING - 1 positions ahead of the compute loop through the same (offset, tile) order
    unsigned long long gk_bits = kbits;
    int gk = -1, gslot = 0, in_flight = 0;
    unsigned grest = 0;
    auto gather_next = [&]() {                          // wave-uniform control
      if (!grest) {
        if (!gk_bits) return;
        gk = __ffsll((long long)gk_bits) - 1;
        gk_bits &= gk_bits - 1;
        grest = (unsigned)__builtin_amdgcn_readlane((int)maskreg, gk);
      }
      const int g = __ffs(grest) - 1;
      grest &= grest - 1;
      gather(gk, g, gslot);
      gslot = gslot + 1 == DW_RING ? 0 : gslot + 1;
      ++in_flight;
    };
#pragma unroll
    for (int i = 0; i < DW_RING - 1; ++i) gather_next();

    int k = __ffsll((long long)kbits) - 1;
    kbits &= kbits - 1;
    unsigned m = (unsigned)__builtin_amdgcn_readlane((int)maskreg, k);
    // weight slabs: B = current offset, Bn = next active offset, requested one offset ahead (16 loads in the same in-order
    // queue as the gathers).  b_pending: Bn's loads may still be in flight; b_after: gathers issued after them.
    f32x4 B[4][4], Bn[4][4];
    load_b(k, B);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // once, before the loop (also lands the first gathers)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) asm volatile("" : "+v"(B[t][s2]));
    int slot = 0, b_after = 0;
    bool b_pending = false;
    while (true) {
      const int kn = kbits ? __ffsll((long long)kbits) - 1 : -1;
      if (kn >= 0) load_b(kn, Bn), b_pending = true, b_after = 0;
#pragma unroll
      for (int g = 0; g < DW_G; ++g) {
        if (!((m >> g) & 1u)) continue;                 // wave-uniform
        {                                               // keep DW_RING - 1 positions in flight behind this one
          const int before = in_flight;
          gather_next();
          b_after += in_flight - before;
        }
        // this position's 4 DMA instructions are the oldest gather in flight: everything issued after them may stay outstanding
        {
          const int newer = in_flight - 1;              // gathers issued after it
          const bool b_newer = b_pending && newer >= b_after;   // the slab request came after it as well
          dm_wait_vmcnt_x4(newer + (b_newer ? 4 : 0));
        }
        --in_flight;
        const uint32_t ab = ring_base + ((uint32_t)slot << 12);
        f32x4 A0 = dm_read_b128(ab + o0), A1 = dm_read_b128(ab + o1), A2 = dm_read_b128(ab + o2), A3 = dm_read_b128(ab + o3);
        DM_WAIT_LGKM4(A0, A1, A2, A3);
        const f32x4 A[4] = {A0, A1, A2, A3};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s].x, B[t][s].x, acc[g][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s].y, B[t][s].y, acc[g][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s].z, B[t][s].z, acc[g][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s].w, B[t][s].w, acc[g][t], 0, 0, 0);
        }
        slot = slot + 1 == DW_RING ? 0 : slot + 1;
      }
      if (kn < 0) break;
      // next offset: its slab must have landed = at most the gathers issued after the request are outstanding
      dm_wait_vmcnt_x4(b_after < in_flight ? b_after : in_flight);
      b_pending = false;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          asm volatile("" : "+v"(Bn[t][s2]));
          B[t][s2] = Bn[t][s2];
        }
      k = kn;
      kbits &= kbits - 1;
      m = (unsigned)__builtin_amdgcn_readlane((int)maskreg, k);
    }
  }

  // D layout (16x16): col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int g = 0; g < DW_G; ++g)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = tile_row0(g) + q * 4 + r;
        const int col = t * 16 + li;
        if (row < a.n_rows) a.Y[row * 64 + col] = conv_epilogue(acc[g][t][r], col, row, a);
      }
}

static int try_launch_dma64(const ConvArgs& a, hipStream_t st) {
  if (a.Kd != 64 || a.Nc != 64 || a.K > DW_KMAX) return -1;
  const int64_t n_tiles = (a.n_rows + 15) / 16;
  const int64_t n_waves = (n_tiles + DW_G - 1) / DW_G;
  hipLaunchKernelGGL(k_spconv_dmaw64, dim3((unsigned)((n_waves + 3) / 4)), dim3(256), 0, st, a, n_waves);
  return 0;
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

extern "C" int sv_sparse_conv_gather_gemm(const float* X, int64_t n_src, const int32_t* nbr, const float* Wt, float* Y,
                                          int64_t n_rows, int K, int Kd, int Nc, const float* bias, const float* scale, const float* shift,
                                          const float* residual, int relu, void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && K > 0 && Kd > 0 && Nc > 0, "sparse_conv: bad sizes");
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(X && nbr && Wt && Y, "sparse_conv: null pointer");
  SV_CHECK_ARG((scale == nullptr) == (shift == nullptr), "sparse_conv: scale and shift go together");
  ConvArgs a{X, nbr, Wt, Y, bias, scale, shift, residual, relu, n_rows, K, Kd, Nc};
  hipStream_t st = sv_stream(stream);
  const int nt = Nc / 16;
  const bool mfma_ok = (Kd % 16 == 0) && (Nc % 16 == 0) && (nt == 1 || nt == 2 || nt == 4 || nt == 8) &&
                       ((uintptr_t)X % 16 == 0) && ((uintptr_t)Wt % 16 == 0);
  static const bool force_v1 = getenv("SEEVCN_SPCONV_V1") != nullptr;
  static const bool no_dma = getenv("SEEVCN_SPCONV_NODMA") != nullptr;
  if (mfma_ok && !force_v1 && !no_dma && try_launch_dma64(a, st) == 0) {
    SV_LAUNCH_CHECK();
    return SV_OK;
  }
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

struct WgradArgs;
static int wgrad_chunk_rows(int64_t n_rows, int K, int groups) {
  // aim at ~2048 workgroups: chunk = n_rows*K*groups/2048 rounded up to 256 rows, within [256, WG_CHUNK_MAX]
  int64_t c = (n_rows * K * groups + 2047) / 2048;
  c = (c + 255) / 256 * 256;
  if (c < 256) c = 256;
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

// grid = (nchunks, K, tile groups).  Each wave walks its share of the chunk 64 rows at a time: one coalesced read of the
// neighbour table, ballot + prefix popcount compaction of the valid (row, source) pairs into a wave-private LDS list,
// then MFMAs over the COMPACTED pairs only (4 pairs per 16x16x4 step) with the next step's operands requested first.
// The four waves' accumulators are summed through LDS in a fixed order and one slab per (chunk, k) is stored.
template <int CT, int NTL>  // register tile grid: CT x NTL tiles of 16x16 (rows = c_in, cols = c_out)
__global__ __launch_bounds__(256) void k_spconv_wgrad(WgradArgs a) {
  __shared__ int32_t pj[4][64], pr[4][64];
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

  auto load_ops = [&](int p, int cnt, float (&xa)[CT], float (&yb)[NTL]) {
    const bool ok = p < cnt;
    const int32_t j = ok ? pj[wid][p] : 0;
    const int32_t r = ok ? pr[wid][p] : 0;
#pragma unroll
    for (int c = 0; c < CT; ++c) xa[c] = (ok && c_base + c * 16 + li < a.Cin) ? a.X[(int64_t)j * a.Cin + c_base + c * 16 + li] : 0.f;
#pragma unroll
    for (int t = 0; t < NTL; ++t) yb[t] = ok ? a.dY[((int64_t)r_begin + r) * a.Cout + n_base + t * 16 + li] : 0.f;
  };

  for (int64_t base = r_begin + wid * 64; base < r_end; base += 256) {
    const int64_t r = base + lane;
    const int32_t j = r < r_end ? nb[r] : -1;
    const unsigned long long m = __ballot(j >= 0);
    const int cnt = __popcll(m);
    if (cnt == 0) continue;
    if (j >= 0) {
      const int pos = __popcll(m & ((1ull << lane) - 1ull));
      pj[wid][pos] = j;
      pr[wid][pos] = (int32_t)(r - r_begin);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-private list: LDS ops of one wave complete in order
    float xa[CT], yb[NTL], xn[CT], yn[NTL];
    load_ops(kk, cnt, xa, yb);
    for (int g = 0; g < cnt; g += 4) {
      if (g + 4 < cnt) load_ops(g + 4 + kk, cnt, xn, yn);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above the MFMA block
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[c], yb[t], acc[c][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 4 < cnt) {
#pragma unroll
        for (int c = 0; c < CT; ++c) xa[c] = xn[c];
#pragma unroll
        for (int t = 0; t < NTL; ++t) yb[t] = yn[t];
      }
    }
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
  // D layout: col = lane&15 (c_out), row = 4*(lane>>4) + reg (c_in)
  float* out = a.partial + (((int64_t)chunk * a.K + k) * a.Cin) * a.Cout;
  for (int e = tid; e < CT * NTL * 256; e += 256) {
    const int ln = e & 63, r = (e >> 6) & 3, tile = e >> 8;
    const int c = tile / NTL, t = tile - c * NTL;
    const int crow = c_base + c * 16 + (ln >> 4) * 4 + r;
    if (crow < a.Cin) out[(int64_t)crow * a.Cout + n_base + t * 16 + (ln & 15)] = red[e];
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
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(sv_grid_1d(slab, 256)), dim3(256), 0, st, a.partial, nslabs, slab, dW);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
