// VCN surface-completion network kernels (see/surface_completion/models/vcn/models/VCN_VC.py:178-214).
//
// The network is a chain of per-point 1x1 convolutions (= GEMMs over M = B*n points) with max-pools over
// the n points of each object, plus a few per-object 3x3 transforms.  It is the one MFMA-bound stage of
// the hot path: 0.99 GMAC/object in exact fp32.  gfx950 has an fp32-input MFMA (v_mfma_f32_32x32x2_f32,
// 64 FLOP/clk/SIMD = the fp32 vector peak, 157 TFLOP/s) whose result is bit-for-bit an fmaf chain, so the
// GEMMs run on the matrix cores in fp32 and the VALU stays free for the epilogues.
//
// Layout: activations are channel-last (M, C) fp32 (the reference's (B,C,n) Conv1d layout transposed),
// weights keep PyTorch's (C_out, C_in) layout, so both GEMM operands are K-contiguous.
#include <math.h>

#include <algorithm>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ACT_RELU = SV_ACT_RELU, ACT_LRELU = SV_ACT_LRELU;
constexpr int EPI_STORE = 1, EPI_MAX = 2, EPI_PARTIAL = 4;      // EPI_PARTIAL: split K, raw sums of blockIdx.z's K range to C[z][M][N]

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_LRELU) return v >= 0.f ? v : v * slope;
  return v;
}

// order-independent float max through integer atomics; *addr must start at -inf
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.f)
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// ------------------------------------------------------------------------------------------------
// C[M,N] = act(A[M,K] * W[N,K]^T + bias[N] + group_bias[row / rows_per_group][N])
//   EPI_STORE: write C;  EPI_MAX: column max over each group of rows_per_group rows -> gmax[group][N]
// 128x128 tile, K tile BK (16), 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles, double-buffered LDS with
// a BK + 4-float row pitch, register-staged global loads.
// K order inside an 8-group is permuted identically for A and W (lane half h takes k = 8q+4h+t).
// ------------------------------------------------------------------------------------------------
// K tile of the LDS stages and workgroups per CU.  Round 5: 16 / 3 (40 KB of LDS, 124 VGPRs: three to four workgroups per CU) instead of 32 / 2
// (74 KB, 156 VGPRs: two) -- stage A 0.759 -> 0.74 ms, same box (tools/build_variant.sh ... vcn); twice the barriers, half again the waves to cover them.
#ifndef SEEVCN_GEMM_BK
#define SEEVCN_GEMM_BK 16
#endif
#ifndef SEEVCN_GEMM_WGS
#define SEEVCN_GEMM_WGS 4
#endif
constexpr int BM = 128, BN = 128, BK = SEEVCN_GEMM_BK, LDP = BK + 4;     // K tile of the LDS stages (callers keep K % 32 == 0)
constexpr int GEMM_ROW_THREADS = BK / 4, GEMM_ROWS_PER_PASS = 256 / GEMM_ROW_THREADS;

struct GemmArgs {
  const float* A; int lda;
  const float* W; int ldw;
  const float* bias;        // [N] or null
  const float* group_bias;  // [M / rows_per_group][N] or null
  int rows_per_group;
  const int32_t* row_group; // [M] group of each row (non-decreasing) or null: row / rows_per_group
  float* C; int ldc;        // EPI_STORE
  float* gmax;              // EPI_MAX: [M / rows_per_group][N], pre-filled with -inf
  int M, N, K;
  int act; float slope;
  const int32_t* m_dev;     // optional: the true number of rows lives on the device (<= M, which then sizes the grid and the buffers)
  int k_chunk;              // EPI_PARTIAL: K range of a split (a multiple of BK); workgroup z takes [z * k_chunk, min(K, (z + 1) * k_chunk))
  int n_tiles;              // column tiles of the launch (the grid is 1-D over row-block-major tiles when tile_mode != 0)
  int tile_mode;            // 0: 128 x 128 tiles on a (N tiles, M tiles[, splits]) grid; 1: 64 x 128; 2: 64 x 64; -1: chosen by the kernel from *m_dev
};

// The tile body: 4 waves (2 x 2), each (32 MT) x (32 NT) of a (64 MT) x (64 NT) tile.  MT = NT = 2 is the 128 x 128 tile; the smaller ones exist for
// products with few rows: VCN's distinct rows are ~15 k of a 65 536-row capacity, i.e. 115 row blocks of 128 -- 460 / 230 / 115 workgroups for
// N = 512 / 256 / 128 on 256 CUs that hold three each (0.41 / 0.20 / 0.05 of the fp32 MFMA peak, tools/vcn_gemm_trace.py).  Every output element sums
// its K products in the same order whatever the tile shape: the results are bit-identical.
template <int EPI, int MT, int NT>
__device__ __forceinline__ void gemm_tile(GemmArgs& g, int m0, int n0, float (*As)[BM * LDP], float (*Bs)[BN * LDP]) {
  constexpr int TM = 64 * MT, TN = 64 * NT, PA = TM / GEMM_ROWS_PER_PASS, PB = TN / GEMM_ROWS_PER_PASS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = tid / GEMM_ROW_THREADS, lc = (tid % GEMM_ROW_THREADS) * 4;  // staging: row lr (+GEMM_ROWS_PER_PASS*i), k offset lc

  float4 ra[PA], rb[PB];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int m = m0 + lr + GEMM_ROWS_PER_PASS * i;
      ra[i] = (m < g.M) ? *reinterpret_cast<const float4*>(g.A + (int64_t)m * g.lda + k0 + lc) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int n = n0 + lr + GEMM_ROWS_PER_PASS * i;
      rb[i] = (n < g.N) ? *reinterpret_cast<const float4*>(g.W + (int64_t)n * g.ldw + k0 + lc) : make_float4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<float4*>(&As[buf][(lr + GEMM_ROWS_PER_PASS * i) * LDP + lc]) = ra[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<float4*>(&Bs[buf][(lr + GEMM_ROWS_PER_PASS * i) * LDP + lc]) = rb[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int k_begin = (EPI & EPI_PARTIAL) ? (int)blockIdx.z * g.k_chunk : 0;
  const int nk = ((EPI & EPI_PARTIAL) ? min(g.K - k_begin, g.k_chunk) : g.K) / BK;
  gload(k_begin);
  lstore(0);
  __syncthreads();
  const int arow = wm * 32 * MT + (lane & 31), brow = wn * 32 * NT + (lane & 31), kh = 4 * (lane >> 5);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(k_begin + (kt + 1) * BK);
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float4 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float4*>(&As[buf][(arow + 32 * i) * LDP + kg * 8 + kh]);
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float4*>(&Bs[buf][(brow + 32 * j) * LDP + kg * 8 + kh]);
      // the accumulators in turn: consecutive MFMAs never share one (written tile by tile, hipcc alternated two of them: every MFMA waited
      // for the one before the last)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int h = lane >> 5;
  if constexpr ((EPI & EPI_PARTIAL) == 0) {
    // the common tile -- every row and column inside the matrix, every row of ONE group (rows of a group are contiguous) -- without the per-element
    // tests, group look-ups and exec-mask branches of the general form below (64 outputs per lane: they were a third of a K = 256 tile's time).
    // Same expression per element, (acc + bias) + group bias -> activation: bit-identical.
    bool fast = m0 + TM <= g.M && n0 + TN <= g.N;
    int g0 = 0, g1 = 0, split = TM;                             // rows [0, split) of the tile belong to g0, [split, TM) to g1
    if (fast && (g.group_bias || (EPI & EPI_MAX))) {
      const int last = m0 + TM - 1;
      g0 = g.row_group ? g.row_group[m0] : m0 / g.rows_per_group;
      g1 = g.row_group ? g.row_group[last] : last / g.rows_per_group;
      if (g1 != g0) {
        // two groups (an object's rows end inside the tile: every other 128-row tile of a batch of ~230-row objects): where does the second
        // start?  Groups are non-decreasing, so the tile holds exactly two iff the first row that is not g0's is g1's.
        if (g.row_group) {
          const bool da = g.row_group[m0 + lane] != g0, db = TM > 64 ? g.row_group[m0 + (TM > 64 ? 64 : 0) + lane] != g0 : false;
          const unsigned long long ba = __ballot(da), bb = __ballot(db);
          split = ba ? __builtin_ctzll(ba) : 64 + (bb ? __builtin_ctzll(bb) : 63);
          fast = g.row_group[m0 + split] == g1;
        } else {
          split = (g0 + 1) * g.rows_per_group - m0;
          fast = g1 == g0 + 1;
        }
      }
    }
    if (fast) {
      const bool two = g1 != g0;                                // uniform
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * 32 * NT + j * 32 + (lane & 31);
        const float bv = g.bias ? g.bias[col] : 0.f;
        const float gv0 = g.group_bias ? g.group_bias[(int64_t)g0 * g.N + col] : 0.f;
        float* out = (EPI & EPI_STORE) ? g.C + (int64_t)(m0 + wm * 32 * MT + 4 * h) * g.ldc + col : nullptr;
        if (!two) {
          float cmax = -INFINITY;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = acc[i][j][r] + bv;
              if (g.group_bias) v += gv0;
              v = apply_act(v, g.act, g.slope);
              if (EPI & EPI_STORE) out[(int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc] = v;
              if (EPI & EPI_MAX) cmax = fmaxf(cmax, v);
            }
          }
          if (EPI & EPI_MAX) {
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
            if (h == 0) atomic_max_f32(&g.gmax[(int64_t)g0 * g.N + col], cmax);
          }
        } else {
          const float gv1 = g.group_bias ? g.group_bias[(int64_t)g1 * g.N + col] : 0.f;
          const int rel0 = wm * 32 * MT + 4 * h;                // tile-relative row of (i = 0, r = 0)
          float cmax0 = -INFINITY, cmax1 = -INFINITY;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool second = rel0 + i * 32 + (r & 3) + 8 * (r >> 2) >= split;
              float v = acc[i][j][r] + bv;
              if (g.group_bias) v += second ? gv1 : gv0;
              v = apply_act(v, g.act, g.slope);
              if (EPI & EPI_STORE) out[(int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc] = v;
              if (EPI & EPI_MAX) {
                cmax0 = fmaxf(cmax0, second ? -INFINITY : v);
                cmax1 = fmaxf(cmax1, second ? v : -INFINITY);
              }
            }
          }
          if (EPI & EPI_MAX) {
            cmax0 = fmaxf(cmax0, __shfl_xor(cmax0, 32, 64));
            cmax1 = fmaxf(cmax1, __shfl_xor(cmax1, 32, 64));
            if (h == 0 && cmax0 > -INFINITY) atomic_max_f32(&g.gmax[(int64_t)g0 * g.N + col], cmax0);
            if (h == 0 && cmax1 > -INFINITY) atomic_max_f32(&g.gmax[(int64_t)g1 * g.N + col], cmax1);
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * 32 * NT + j * 32 + (lane & 31);
    const bool cok = col < g.N;
    const float bv = (cok && g.bias) ? g.bias[col] : 0.f;
    float cmax = -INFINITY;
    int cgroup = -1;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 * MT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (!cok || row >= g.M) continue;
        if constexpr ((EPI & EPI_PARTIAL) != 0) {
          g.C[((int64_t)blockIdx.z * g.M + row) * g.ldc + col] = acc[i][j][r];
          continue;
        }
        const int grp = g.row_group ? g.row_group[row] : row / g.rows_per_group;
        float v = acc[i][j][r] + bv;
        if (g.group_bias) v += g.group_bias[(int64_t)grp * g.N + col];
        v = apply_act(v, g.act, g.slope);
        if (EPI & EPI_STORE) g.C[(int64_t)row * g.ldc + col] = v;
        if (EPI & EPI_MAX) {
          if (grp != cgroup) {
            if (cgroup >= 0) atomic_max_f32(&g.gmax[(int64_t)cgroup * g.N + col], cmax);
            cgroup = grp;
            cmax = v;
          } else {
            cmax = fmaxf(cmax, v);
          }
        }
      }
    }
    if (EPI & EPI_MAX) {
      // combine the two lane halves when they ended in the same group (the common case: tile inside one object)
      const float omax = __shfl_xor(cmax, 32, 64);
      const int ogroup = __shfl_xor(cgroup, 32, 64);
      if (cgroup >= 0) {
        if (ogroup == cgroup) {
          if (h == 0) atomic_max_f32(&g.gmax[(int64_t)cgroup * g.N + col], fmaxf(cmax, omax));
        } else {
          atomic_max_f32(&g.gmax[(int64_t)cgroup * g.N + col], cmax);
        }
      }
    }
  }
}

// tile shape for (rows, N): the largest tile that still gives the chip >= GEMM_FILL workgroups (256 CUs x 3 resident), 128 x 128 when none does not
constexpr int GEMM_FILL = 600;
__host__ __device__ inline int gemm_tile_mode(int rows, int N) {
  const int t128 = ((rows + 127) / 128) * ((N + 127) / 128);
  if (t128 >= GEMM_FILL) return 0;
  if (2 * t128 >= GEMM_FILL || N <= 64) return 1;              // 64 x 128
  return 2;                                                     // 64 x 64
}
__host__ __device__ inline int gemm_tiles(int rows, int N, int mode) {
  return mode == 0 ? ((rows + 127) / 128) * ((N + 127) / 128) : mode == 1 ? ((rows + 63) / 64) * ((N + 127) / 128) : ((rows + 63) / 64) * ((N + 63) / 64);
}

template <int EPI>
__global__ __launch_bounds__(256, SEEVCN_GEMM_WGS) void k_gemm_f32(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[2][BM * LDP];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDP];
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, so give each XCD a contiguous run of the
  // (row-block major) tile list: the N-tiles that share an A row-block then hit the same L2 instead of 8 different ones.
  int lin = blockIdx.y * gridDim.x + blockIdx.x;
  int mode = g.tile_mode;
  if (g.m_dev) {
    // rows counted on the device (VCN's distinct rows: no host read): the launch is sized for the capacity (and for the finest tiling), the tile
    // shape is chosen here from the real count, the workgroups past the last real tile leave at once, and the XCD-aware order is made over the real
    // tiles -- the first `total` workgroup ids, dealt to the XCDs round-robin like any launch
    g.M = min(*g.m_dev, g.M);
    if (mode < 0) mode = gemm_tile_mode(g.M, g.N);
  }
  const int n_tiles = (EPI & EPI_PARTIAL) ? (int)gridDim.x : mode <= 1 ? (g.N + 127) / 128 : (g.N + 63) / 64;
  const int total = (EPI & EPI_PARTIAL) ? (int)(gridDim.x * gridDim.y) : gemm_tiles(g.M, g.N, mode);
  if (lin >= total) return;
  if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
  const int mt = lin / n_tiles, nt = lin % n_tiles;
  if constexpr ((EPI & EPI_PARTIAL) != 0) {
    gemm_tile<EPI, 2, 2>(g, mt * 128, nt * 128, As, Bs);
  } else {
    if (mode == 0) gemm_tile<EPI, 2, 2>(g, mt * 128, nt * 128, As, Bs);
    else if (mode == 1) gemm_tile<EPI, 1, 2>(g, mt * 64, nt * 128, As, Bs);
    else gemm_tile<EPI, 1, 1>(g, mt * 64, nt * 64, As, Bs);
  }
}

__global__ void k_fill_f32(float* p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

extern "C" int sv_fill_f32(float* dst, int64_t n, float value, void* stream) {
  if (n <= 0) return SV_OK;
  SV_CHECK_ARG(dst, "fill_f32: null pointer");
  hipLaunchKernelGGL(k_fill_f32, dim3(sv_grid_1d(n, 256)), dim3(256), 0, sv_stream(stream), dst, n, value);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// M <= 64 rows (the per-object layers: pose_fc, shape_fc, the folded global half of mlp_conv2[0]; VCN_VC.py:124-131,200-204).
// These are weight-streaming: 2*N*K*4 bytes against 2*M*N*K flops.  The 128x128 tile kernel gives them N/128 workgroups that
// each walk all of K alone (85 us for a 4 MB weight matrix).  Here a workgroup owns 16 output columns and all rows; its four
// waves share K in interleaved 16-wide steps (16x16x4 MFMA, one 16-byte load per operand and 16 k), partial sums meet in LDS in a fixed order.
// ------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
// NW waves share K (4, or 16 from K = 512 on: a wave then has at most SM_U steps and requests everything it will ever read in one go -- with 4 waves
// a 1024-long K is four dependent rounds of load latency, 10.6 us for a layer whose 4 MB of weights stream in 1.3)
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_gemm_small_m(GemmArgs g) {
  __shared__ float s_red[NW][4][4][64];                     // [wave][row tile][reg][lane]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const int n0 = blockIdx.x * 16;
  // K % 16 == 0 (checked by the launcher): the waves take the 16-wide k-steps round-robin
  f32x4v acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  const int wn = n0 + li;
  const float* wrow = g.W + (int64_t)(wn < g.N ? wn : g.N - 1) * g.ldw + 4 * kk;
  const float* arow[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int m = t * 16 + li;
    arow[t] = g.A + (int64_t)(m < g.M ? m : g.M - 1) * g.lda + 4 * kk;
  }
  // lane (li, kk) holds k = k0 + 4*kk + s in MFMA s, for A and W alike.  SM_U k-steps of a wave are requested together: one step at a time the
  // kernel was a chain of load latencies (13-16 us for a 4 MB weight matrix = 0.3 TB/s on 64 workgroups)
  constexpr int SM_U = 4, KS = 16 * NW;                       // KS: k covered by one step of all waves
  int k0 = wid * 16;
  for (; k0 + KS * (SM_U - 1) < g.K; k0 += KS * SM_U) {
    f32x4v b[SM_U], a[SM_U][4];
#pragma unroll
    for (int u = 0; u < SM_U; ++u) {
      b[u] = *reinterpret_cast<const f32x4v*>(wrow + k0 + KS * u);
#pragma unroll
      for (int t = 0; t < 4; ++t) a[u][t] = *reinterpret_cast<const f32x4v*>(arow[t] + k0 + KS * u);
    }
#pragma unroll
    for (int u = 0; u < SM_U; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t][s], b[u][s], acc[t], 0, 0, 0);
  }
  for (; k0 < g.K; k0 += KS) {
    const f32x4v b = *reinterpret_cast<const f32x4v*>(wrow + k0);
    f32x4v a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const f32x4v*>(arow[t] + k0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s], b[s], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_red[wid][t][r][lane] = acc[t][r];
  __syncthreads();
  // D layout (16x16): col = lane & 15, row = 4 * (lane >> 4) + reg.  Thread (wid = row tile, lane) finishes 4 outputs.
  const int col = n0 + li;
  if (col >= g.N || wid >= 4) return;
  const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = wid * 16 + 4 * kk + r;
    if (row >= g.M) continue;
    float v = 0.f;                                            // partials of groups of four waves, pairwise, in wave order: a fixed order
#pragma unroll
    for (int q = 0; q < NW; q += 4)
      v += (s_red[q][wid][r][lane] + s_red[q + 1][wid][r][lane]) + (s_red[q + 2][wid][r][lane] + s_red[q + 3][wid][r][lane]);
    g.C[(int64_t)row * g.ldc + col] = apply_act(v + bv, g.act, g.slope);
  }
}

// ---- split K: a product with few output tiles and a long K (PV-RCNN's shared FC over the pooled RoI grid, roi_head_template / pvrcnn_head.py:
// 512 RoIs x 27 648 -> 256 = 8 tiles walking K alone: 1.9 ms on 8 CUs) runs its K ranges on blockIdx.z and the raw sums meet in a second pass that
// adds them in split order (fixed: bitwise reproducible), then bias, group bias and activation as the one-pass epilogue does.
__global__ __launch_bounds__(256) void k_gemm_split_finish(const float* __restrict__ partial, int splits, GemmArgs g) {
  const int64_t total = (int64_t)g.M * g.N;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int row = (int)(e / g.N), col = (int)(e % g.N);
    float v = 0.f;
    for (int p = 0; p < splits; ++p) v += partial[(int64_t)p * total + e];
    if (g.bias) v += g.bias[col];
    if (g.group_bias) v += g.group_bias[(int64_t)(row / g.rows_per_group) * g.N + col];
    g.C[(int64_t)row * g.ldc + col] = apply_act(v, g.act, g.slope);
  }
}

// splits the split-K form would use for this product (1 = it does not apply): fewer than 64 output tiles and at least 16 K-steps per split
extern "C" int sv_gemm_splitk_splits(int M, int N, int K) {
  if (M <= 64 || K % BK != 0) return 1;
  const int tiles = sv_div_up(M, BM) * sv_div_up(N, BN);
  if (tiles >= 64) return 1;
  int s = sv_div_up(512, tiles);
  const int most = K / (16 * BK);
  if (s > most) s = most;
  if (s < 2) return 1;
  const int chunk = sv_div_up(sv_div_up(K, s), BK) * BK;
  return sv_div_up(K, chunk);
}
extern "C" size_t sv_gemm_splitk_scratch_bytes(int M, int N, int K) {
  const int s = sv_gemm_splitk_splits(M, N, K);
  return s > 1 ? (size_t)s * M * N * sizeof(float) : 256;
}

static int gemm_launch(const float* A, int lda, const float* W, int ldw, const float* bias, const float* group_bias, int rows_per_group,
                       const int32_t* row_group, float* C, int ldc, float* group_max, int M, int N, int K, int act, float slope,
                       void* stream, const int32_t* m_dev = nullptr) {
  SV_CHECK_ARG(A && W && (C || group_max), "gemm_bias_act: null pointer");
  SV_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % BK == 0, "gemm_bias_act: K=%d must be a positive multiple of %d", K, BK);
  SV_CHECK_ARG(lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldw >= K, "gemm_bias_act: lda/ldw must be >= K and multiples of 4");
  SV_CHECK_ARG(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0), "gemm_bias_act: A/W must be 16-byte aligned");
  SV_CHECK_ARG(rows_per_group >= 1, "gemm_bias_act: rows_per_group must be >= 1");
  SV_CHECK_ARG(act >= 0 && act <= 2, "gemm_bias_act: unknown activation %d", act);
  GemmArgs g{A, lda, W, ldw, bias, group_bias, rows_per_group, row_group, C, ldc, group_max, M, N, K, act, slope, m_dev, 0, 0, 0};
  // tile shape: from M when the host knows it; with the row count on the device the kernel chooses (mode -1) and the 1-D grid covers the finest
  // tiling of the capacity.  SEEVCN_GEMM_TILES=0: 128 x 128 always (A/B runs)
  static const bool big_only = getenv("SEEVCN_GEMM_TILES") && atoi(getenv("SEEVCN_GEMM_TILES")) == 0;
  g.tile_mode = big_only ? 0 : (m_dev ? -1 : gemm_tile_mode(M, N));
  const int wgs = g.tile_mode < 0 ? std::max(std::max(gemm_tiles(M, N, 0), gemm_tiles(M, N, 1)), gemm_tiles(M, N, 2)) : gemm_tiles(M, N, g.tile_mode);
  dim3 grid(wgs);
  hipStream_t st = sv_stream(stream);
  if (M <= 64 && C && !group_max && !group_bias && !row_group && !m_dev) {
    static const int small_waves = getenv("SEEVCN_GEMM_SMALL_WAVES") ? atoi(getenv("SEEVCN_GEMM_SMALL_WAVES")) : 0;      // 4 / 16: A/B runs
    if (small_waves == 16 || (small_waves != 4 && K >= 512)) hipLaunchKernelGGL(k_gemm_small_m<16>, dim3(sv_div_up(N, 16)), dim3(1024), 0, st, g);
    else hipLaunchKernelGGL(k_gemm_small_m<4>, dim3(sv_div_up(N, 16)), dim3(256), 0, st, g);
    SV_LAUNCH_CHECK();
    return SV_OK;
  }
  if (C && group_max)
    hipLaunchKernelGGL(k_gemm_f32<EPI_STORE | EPI_MAX>, grid, dim3(256), 0, st, g);
  else if (C)
    hipLaunchKernelGGL(k_gemm_f32<EPI_STORE>, grid, dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL(k_gemm_f32<EPI_MAX>, grid, dim3(256), 0, st, g);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_gemm_bias_act(const float* A, int lda, const float* W, int ldw, const float* bias,
                                const float* group_bias, int rows_per_group, float* C, int ldc, float* group_max,
                                int M, int N, int K, int act, float slope, void* stream) {
  return gemm_launch(A, lda, W, ldw, bias, group_bias, rows_per_group, nullptr, C, ldc, group_max, M, N, K, act, slope, stream);
}

// sv_gemm_bias_act (uniform groups, no column max) in split-K form: scratch = sv_gemm_splitk_scratch_bytes(M, N, K); sv_gemm_splitk_splits == 1 runs
// the one-pass kernel.  Same sums in another order: equal to the one-pass result up to fp32 rounding.
extern "C" int sv_gemm_bias_act_splitk(const float* A, int lda, const float* W, int ldw, const float* bias, const float* group_bias, int rows_per_group,
                                       float* C, int ldc, int M, int N, int K, int act, float slope, void* scratch, void* stream) {
  const int splits = (M > 0 && N > 0 && K > 0) ? sv_gemm_splitk_splits(M, N, K) : 1;
  if (splits <= 1) return gemm_launch(A, lda, W, ldw, bias, group_bias, rows_per_group, nullptr, C, ldc, nullptr, M, N, K, act, slope, stream);
  SV_CHECK_ARG(A && W && C && scratch, "gemm_bias_act_splitk: null pointer");
  SV_CHECK_ARG(lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldw >= K, "gemm_bias_act_splitk: lda/ldw must be >= K and multiples of 4");
  SV_CHECK_ARG(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0), "gemm_bias_act_splitk: A/W must be 16-byte aligned");
  SV_CHECK_ARG(rows_per_group >= 1 && act >= 0 && act <= 2 && ldc >= N, "gemm_bias_act_splitk: bad arguments");
  const int chunk = sv_div_up(sv_div_up(K, splits), BK) * BK;
  GemmArgs g{A, lda, W, ldw, bias, group_bias, rows_per_group, nullptr, reinterpret_cast<float*>(scratch), N, nullptr, M, N, K, act, slope, nullptr, chunk, 0, 0};
  hipStream_t st = sv_stream(stream);
  hipLaunchKernelGGL(k_gemm_f32<EPI_PARTIAL>, dim3(sv_div_up(N, BN), sv_div_up(M, BM), splits), dim3(256), 0, st, g);
  g.C = C, g.ldc = ldc;
  hipLaunchKernelGGL(k_gemm_split_finish, dim3(sv_grid_1d((int64_t)M * N, 256)), dim3(256), 0, st, reinterpret_cast<const float*>(scratch), splits, g);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_gemm_bias_act_ragged(const float* A, int lda, const float* W, int ldw, const float* bias, const float* group_bias,
                                       const int32_t* row_group, float* C, int ldc, float* group_max, int M, int N, int K, int act,
                                       float slope, void* stream) {
  SV_CHECK_ARG(row_group, "gemm_bias_act_ragged: row_group is required");
  return gemm_launch(A, lda, W, ldw, bias, group_bias, 1, row_group, C, ldc, group_max, M, N, K, act, slope, stream);
}

// The ragged GEMM with the number of rows on the device: M_cap rows of A / C / row_group are addressable, the first *m_dev are computed (the
// distinct rows of VCN's input clouds, sv_unique_rows_compact's total -- which the host then never has to read).
extern "C" int sv_gemm_bias_act_ragged_dev(const float* A, int lda, const float* W, int ldw, const float* bias, const float* group_bias,
                                           const int32_t* row_group, float* C, int ldc, float* group_max, int M_cap, const int32_t* m_dev, int N, int K,
                                           int act, float slope, void* stream) {
  SV_CHECK_ARG(row_group && m_dev, "gemm_bias_act_ragged_dev: row_group and m_dev are required");
  return gemm_launch(A, lda, W, ldw, bias, group_bias, 1, row_group, C, ldc, group_max, M_cap, N, K, act, slope, stream, m_dev);
}

// ------------------------------------------------------------------------------------------------
// out[m][c] = act(w[c][0]*x + w[c][1]*y + w[c][2]*z + b[c]) : the K=3 first layers (VALU, HBM-write-bound)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pointwise3(const float* __restrict__ xyz, const float* __restrict__ w,
                                                    const float* __restrict__ b, float* __restrict__ out, int64_t M,
                                                    int C, int act, float slope, const int64_t* __restrict__ sel, const int32_t* __restrict__ m_dev) {
  const int cq = C / 4;
  if (m_dev) M = min((int64_t)*m_dev, M);
  const int64_t total = M * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / cq;
    const int c = (int)(i - m * cq) * 4;
    const int64_t src = sel ? sel[m] : m;                       // optional row gather (the distinct rows of the clouds)
    const float x = xyz[src * 3], y = xyz[src * 3 + 1], z = xyz[src * 3 + 2];
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* wr = w + (c + j) * 3;
      // same association as a K=3 dot product accumulated left to right, then bias
      float v = fmaf(wr[2], z, fmaf(wr[1], y, wr[0] * x));
      v += b ? b[c + j] : 0.f;
      o[j] = apply_act(v, act, slope);
    }
    *reinterpret_cast<float4*>(out + m * C + c) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int sv_pointwise_conv3(const float* xyz, const float* weight, const float* bias, float* out, int64_t M,
                                  int C, int act, float slope, void* stream) {
  SV_CHECK_ARG(xyz && weight && out && M > 0 && C > 0 && C % 4 == 0, "pointwise_conv3: bad arguments (C must be a multiple of 4)");
  hipLaunchKernelGGL(k_pointwise3, dim3(sv_grid_1d(M * (C / 4), 256)), dim3(256), 0, sv_stream(stream), xyz, weight, bias,
                     out, M, C, act, slope, nullptr, nullptr);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// The same layer on GATHERED rows out[m] = f(xyz[sel[m]]) for m < *m_dev (<= M_cap): the gather and the count stay on the device.
extern "C" int sv_pointwise_conv3_gather(const float* xyz, const int64_t* sel, int64_t M_cap, const int32_t* m_dev, const float* weight, const float* bias,
                                         float* out, int C, int act, float slope, void* stream) {
  SV_CHECK_ARG(xyz && sel && m_dev && weight && out && M_cap > 0 && C > 0 && C % 4 == 0, "pointwise_conv3_gather: bad arguments (C must be a multiple of 4)");
  hipLaunchKernelGGL(k_pointwise3, dim3(sv_grid_1d(M_cap * (C / 4), 256, 2048)), dim3(256), 0, sv_stream(stream), xyz, weight, bias, out, M_cap, C, act, slope,
                     sel, m_dev);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// Per-object geometry (one workgroup per object; n points).  state[b] = 32 floats:
//   [0] frustum angle  [1..3] mean of the frustum-view cloud  [4..6] centre  [7..15] rot (row-major 3x3)
// ------------------------------------------------------------------------------------------------
constexpr int ST = 32;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// VCN_VC.py:185-190: angle = atan2(mean y, mean x); rotate by -angle; subtract the mean of the rotated cloud
__global__ __launch_bounds__(256) void k_vcn_vc_prep(const float* __restrict__ in, int n, float* __restrict__ fview,
                                                     float* __restrict__ centred, float* __restrict__ state) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* p = in + (int64_t)b * n * 3;
  float sx = 0.f, sy = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { sx += p[i * 3]; sy += p[i * 3 + 1]; }
  const float mx = block_sum_256(sx, red) / (float)n;
  const float my = block_sum_256(sy, red) / (float)n;
  const float ang = atan2f(my, mx);
  // rotate_points_along_z(points, -angle) (utils/transform.py:33-57): R = [[c,s,0],[-s,c,0],[0,0,1]], p @ R
  const float c = cosf(-ang), s = sinf(-ang);
  float ax = 0.f, ay = 0.f, az = 0.f;
  float* f = fview + (int64_t)b * n * 3;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
    const float rx = x * c - y * s, ry = x * s + y * c;
    f[i * 3] = rx; f[i * 3 + 1] = ry; f[i * 3 + 2] = z;
    ax += rx; ay += ry; az += z;
  }
  const float m0 = block_sum_256(ax, red) / (float)n;
  const float m1 = block_sum_256(ay, red) / (float)n;
  const float m2 = block_sum_256(az, red) / (float)n;
  float* o = centred + (int64_t)b * n * 3;
  for (int i = threadIdx.x; i < n; i += 256) {
    o[i * 3] = f[i * 3] - m0; o[i * 3 + 1] = f[i * 3 + 1] - m1; o[i * 3 + 2] = f[i * 3 + 2] - m2;
  }
  if (threadIdx.x == 0) {
    float* st = state + (int64_t)b * ST;
    st[0] = ang; st[1] = m0; st[2] = m1; st[3] = m2;
  }
}

// VCN_VC.py:195-200: centre = mean + trans; rot = ortho6d -> R (12-49); pc_cn = (fview - centre) @ R^T
__global__ __launch_bounds__(256) void k_vcn_vc_pose(const float* __restrict__ fview, int n, const float* __restrict__ rel_pose,
                                                     float* __restrict__ state, float* __restrict__ pc_cn) {
  const int b = blockIdx.x;
  float* st = state + (int64_t)b * ST;
  const float* rp = rel_pose + (int64_t)b * 9;
  const float cx = st[1] + rp[0], cy = st[2] + rp[1], cz = st[3] + rp[2];
  // normalize_vector: v / max(|v|, 1e-8)
  float x0 = rp[3], x1 = rp[4], x2 = rp[5];
  const float y0 = rp[6], y1 = rp[7], y2 = rp[8];
  float mag = fmaxf(sqrtf(x0 * x0 + x1 * x1 + x2 * x2), 1e-8f);
  x0 /= mag; x1 /= mag; x2 /= mag;
  float z0 = x1 * y2 - x2 * y1, z1 = x2 * y0 - x0 * y2, z2 = x0 * y1 - x1 * y0;
  mag = fmaxf(sqrtf(z0 * z0 + z1 * z1 + z2 * z2), 1e-8f);
  z0 /= mag; z1 /= mag; z2 /= mag;
  const float w0 = z1 * x2 - z2 * x1, w1 = z2 * x0 - z0 * x2, w2 = z0 * x1 - z1 * x0;  // y = z cross x
  // matrix = cat(x, y, z) as COLUMNS: R[r][0]=x[r], R[r][1]=y[r], R[r][2]=z[r]
  const float R[9] = {x0, w0, z0, x1, w1, z1, x2, w2, z2};
  const float* f = fview + (int64_t)b * n * 3;
  float* o = pc_cn + (int64_t)b * n * 3;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float px = f[i * 3] - cx, py = f[i * 3 + 1] - cy, pz = f[i * 3 + 2] - cz;
    // p @ R^T : out[j] = sum_k p[k] * R[j][k]
    o[i * 3] = px * R[0] + py * R[1] + pz * R[2];
    o[i * 3 + 1] = px * R[3] + py * R[4] + pz * R[5];
    o[i * 3 + 2] = px * R[6] + py * R[7] + pz * R[8];
  }
  if (threadIdx.x == 0) {
    st[4] = cx; st[5] = cy; st[6] = cz;
    for (int k = 0; k < 9; ++k) st[7 + k] = R[k];
  }
}

// VCN_VC.py:205-212: coarse_vc = coarse @ R + centre; rotate back by +angle; reg_rot = R @ Rz(angle); reg_centre
__global__ __launch_bounds__(256) void k_vcn_vc_finish(const float* __restrict__ coarse_cn, int nc, const float* __restrict__ state,
                                                       float* __restrict__ coarse, float* __restrict__ reg_rot,
                                                       float* __restrict__ reg_centre) {
  const int b = blockIdx.x;
  const float* st = state + (int64_t)b * ST;
  const float ang = st[0];
  const float c = cosf(ang), s = sinf(ang);
  const float cx = st[4], cy = st[5], cz = st[6];
  float R[9];
  for (int k = 0; k < 9; ++k) R[k] = st[7 + k];
  const float* p = coarse_cn + (int64_t)b * nc * 3;
  float* o = coarse + (int64_t)b * nc * 3;
  for (int i = threadIdx.x; i < nc; i += 256) {
    const float x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
    // p @ R : out[j] = sum_k p[k] * R[k][j]
    const float vx = x * R[0] + y * R[3] + z * R[6] + cx;
    const float vy = x * R[1] + y * R[4] + z * R[7] + cy;
    const float vz = x * R[2] + y * R[5] + z * R[8] + cz;
    o[i * 3] = vx * c - vy * s; o[i * 3 + 1] = vx * s + vy * c; o[i * 3 + 2] = vz;
  }
  if (threadIdx.x == 0) {
    // rot_from_heading(angle) = [[c,s,0],[-s,c,0],[0,0,1]] (utils/transform.py:6-31); reg_rot = R @ that
    float* rr = reg_rot + (int64_t)b * 9;
    for (int r = 0; r < 3; ++r) {
      rr[r * 3] = R[r * 3] * c - R[r * 3 + 1] * s;
      rr[r * 3 + 1] = R[r * 3] * s + R[r * 3 + 1] * c;
      rr[r * 3 + 2] = R[r * 3 + 2];
    }
    float* rc = reg_centre + (int64_t)b * 3;
    rc[0] = cx * c - cy * s; rc[1] = cx * s + cy * c; rc[2] = cz;
  }
}

extern "C" int sv_vcn_vc_prep(const float* input, int batch, int n, float* fview, float* centred, float* state, void* stream) {
  SV_CHECK_ARG(input && fview && centred && state && batch > 0 && n > 0, "vcn_vc_prep: bad arguments");
  hipLaunchKernelGGL(k_vcn_vc_prep, dim3(batch), dim3(256), 0, sv_stream(stream), input, n, fview, centred, state);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_vcn_vc_pose(const float* fview, int batch, int n, const float* rel_pose, float* state, float* pc_cn, void* stream) {
  SV_CHECK_ARG(fview && rel_pose && state && pc_cn && batch > 0 && n > 0, "vcn_vc_pose: bad arguments");
  hipLaunchKernelGGL(k_vcn_vc_pose, dim3(batch), dim3(256), 0, sv_stream(stream), fview, n, rel_pose, state, pc_cn);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_vcn_vc_finish(const float* coarse_cn, int batch, int num_coarse, const float* state, float* coarse,
                                float* reg_rot, float* reg_centre, void* stream) {
  SV_CHECK_ARG(coarse_cn && state && coarse && reg_rot && reg_centre && batch > 0 && num_coarse > 0, "vcn_vc_finish: bad arguments");
  hipLaunchKernelGGL(k_vcn_vc_finish, dim3(batch), dim3(256), 0, sv_stream(stream), coarse_cn, num_coarse, state, coarse,
                     reg_rot, reg_centre);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// VCN_CN geometry (VCN_CN.py:142-156 with utils/transform.py:91-160):
//   to_cn:   out = rotate_z(p - box.xyz, -box.yaw) / box.dx        (vc_to_cn + normalize_scale)
//   to_vc:   out = rotate_z(p * box.dx, +box.yaw) + box.xyz        (restore_scale + cn_to_vc)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vcn_cn_xform(const float* __restrict__ in, int n, const float* __restrict__ boxes,
                                                      int inverse, float* __restrict__ out) {
  const int b = blockIdx.x;
  const float* bx = boxes + (int64_t)b * 7;
  const float cx = bx[0], cy = bx[1], cz = bx[2], len = bx[3], yaw = bx[6];
  const float a = inverse ? yaw : -yaw;
  const float c = cosf(a), s = sinf(a);
  const float* p = in + (int64_t)b * n * 3;
  float* o = out + (int64_t)b * n * 3;
  for (int i = threadIdx.x; i < n; i += 256) {
    float x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
    if (!inverse) {
      x -= cx; y -= cy; z -= cz;
      o[i * 3] = (x * c - y * s) / len; o[i * 3 + 1] = (x * s + y * c) / len; o[i * 3 + 2] = z / len;
    } else {
      x *= len; y *= len; z *= len;
      o[i * 3] = (x * c - y * s) + cx; o[i * 3 + 1] = (x * s + y * c) + cy; o[i * 3 + 2] = z + cz;
    }
  }
}

extern "C" int sv_vcn_cn_transform(const float* in, int batch, int n, const float* gt_boxes, int inverse, float* out, void* stream) {
  SV_CHECK_ARG(in && gt_boxes && out && batch > 0 && n > 0, "vcn_cn_transform: bad arguments");
  hipLaunchKernelGGL(k_vcn_cn_xform, dim3(batch), dim3(256), 0, sv_stream(stream), in, n, gt_boxes, inverse, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
