// Training-side kernels of the dense per-point / per-RoI layer stacks (Conv1d k=1 / Linear + BatchNorm1d + ReLU + max over points):
// VCN's encoder / pose branch in training mode (see/surface_completion/models/vcn/models/VCN_VC.py:97-106,116-131,178-214 under autograd) and
// PV-RCNN's point head / feature fusion / RoI head (detector3d/pcdet/models/dense_heads/point_head_simple.py, backbones_3d/pfe/
// voxel_set_abstraction.py:168-172, roi_heads/pvrcnn_head.py:171-176).  The reference runs them through cuDNN / cuBLAS; the forward GEMM is
// sv_gemm_bias_act (vcn.hip), here is what a backward pass needs on top:
//
//   sv_gemm_tn       C (N, K) = A^T B over M rows: the weight gradient dW = dY^T X.  The contraction runs over the LONG dimension (M = points),
//                    both operands are read row-major as they lie (no transposed copy of a 65 536 x 1024 activation); 128 x 128 output tiles
//                    on v_mfma_f32_32x32x2_f32, M split over workgroups, partial tiles summed in a fixed order (bitwise reproducible).
//   sv_gemm_strided  any small / odd-shaped product (N = 9, K = 3, ...) with element strides, contraction split + fixed-order sum.
//   sv_column_sums   bias gradient.
//   sv_segment_max / _backward, sv_segment_sum   max over the rows of every object with its arg-max, and the two gradients that come back
//                    through it (scatter to the arg-max rows; sum of an object's rows for the broadcast global feature).
#include "common.h"

typedef float dt_f32x16 __attribute__((ext_vector_type(16)));

static size_t dt_align(size_t x) { return (x + 255) / 256 * 256; }

// ------------------------------------------------------------------------------------------------------------------------------ TN GEMM
constexpr int TN_T = 128;          // output tile (n x k)
constexpr int TN_BM = 32;          // rows of the contraction per step
constexpr int TN_LD = 160;         // floats per staged row: 128 + 32, so that the two lane halves of a ds_read_b32 (rows m and m + 1) hit disjoint banks

struct TnArgs {
  const float* A; int64_t lda;     // (M, N)
  const float* B; int64_t ldb;     // (M, K)
  float* C; int64_t ldc;           // (N, K)  (splits == 1) or partial (splits, N, K) contiguous
  int64_t M;
  int N, K, splits;
  int64_t rows_per_split;
};

__global__ __launch_bounds__(256, 2) void k_gemm_tn(TnArgs g) {
  __shared__ __attribute__((aligned(16))) float As[TN_BM * TN_LD];
  __shared__ __attribute__((aligned(16))) float Bs[TN_BM * TN_LD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid >> 1, wk = wid & 1;                       // wave tile: rows (n) wn*64.., cols (k) wk*64..
  const int n0 = blockIdx.y * TN_T, k0 = blockIdx.x * TN_T, sp = blockIdx.z;
  const int64_t m_begin = (int64_t)sp * g.rows_per_split, m_end = min(g.M, m_begin + g.rows_per_split);
  // staging: thread -> row r = tid >> 3 (0..31), four float4 at columns (tid & 7) * 4 + 32 * q
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  float4 ra[4], rb[4];
  auto gload = [&](int64_t m0) {
    const int64_t m = m0 + sr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cn = n0 + sc + 32 * q, ck = k0 + sc + 32 * q;
      ra[q] = (m < m_end && cn < g.N) ? *reinterpret_cast<const float4*>(g.A + m * g.lda + cn) : make_float4(0, 0, 0, 0);
      rb[q] = (m < m_end && ck < g.K) ? *reinterpret_cast<const float4*>(g.B + m * g.ldb + ck) : make_float4(0, 0, 0, 0);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<float4*>(&As[sr * TN_LD + sc + 32 * q]) = ra[q];
      *reinterpret_cast<float4*>(&Bs[sr * TN_LD + sc + 32 * q]) = rb[q];
    }
  };
  dt_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int li = lane & 31, kh = lane >> 5;
  if (m_begin < m_end) {
    gload(m_begin);
    for (int64_t m0 = m_begin; m0 < m_end; m0 += TN_BM) {
      __syncthreads();                                          // the previous step's reads are done
      lstore();
      __syncthreads();
      if (m0 + TN_BM < m_end) gload(m0 + TN_BM);
#pragma unroll
      for (int s = 0; s < TN_BM / 2; ++s) {
        const float* ar = &As[(2 * s + kh) * TN_LD + wn * 64 + li];
        const float* br = &Bs[(2 * s + kh) * TN_LD + wk * 64 + li];
        const float a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
  }
  // D layout of a 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* out = g.splits > 1 ? g.C + (int64_t)sp * g.N * g.K : g.C;
  const int64_t ldo = g.splits > 1 ? g.K : g.ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = k0 + wk * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < g.N && col < g.K) out[(int64_t)row * ldo + col] = acc[i][j][r];
      }
    }
}

// out[e] = sum over the splits of partial[s][e] in split order (fixed: bitwise reproducible); out at (row, col) with leading dimension ldc
__global__ __launch_bounds__(256) void k_split_reduce(const float* __restrict__ partial, int splits, int rows, int cols, float* __restrict__ out, int64_t ldc) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int p = 0; p < splits; ++p) s += partial[(int64_t)p * total + e];
    out[(e / cols) * ldc + (e % cols)] = s;
  }
}

static int tn_splits(int64_t M, int N, int K) {
  const int tiles = sv_div_up(N, TN_T) * sv_div_up(K, TN_T);
  int64_t s = (1024 + tiles - 1) / tiles;                       // aim at ~1024 workgroups (2 per CU fit)
  const int64_t most = (M + 4 * TN_BM - 1) / (4 * TN_BM);       // at least four steps per split
  if (s > most) s = most;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}

static int sg_splits(int I, int J, int64_t L);
extern "C" size_t sv_gemm_tn_scratch_bytes(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 256;
  const int s = tn_splits(M, N, K), s2 = sg_splits(N, K, M);     // whichever path takes the product
  const int m = s > s2 ? s : s2;
  return dt_align((size_t)(m > 1 ? m : 0) * N * K * sizeof(float)) + 256;
}

// ------------------------------------------------------------------------------------------------------------------------------ generic strided
// C[i][j] = sum_c A[i * sai + c * sac] * B[c * sbc + j * sbj]   (I x J outputs, contraction length L; element strides): one thread per output and
// contraction slice, slices summed in a fixed order by k_split_reduce.  For the small and odd-shaped products of a backward pass.
struct SgArgs {
  const float* A; int64_t sai, sac;
  const float* B; int64_t sbc, sbj;
  float* C; int64_t ldc;           // or partial (splits, I, J)
  int I, J, splits;
  int64_t L, len_per_split;
};

__global__ __launch_bounds__(256) void k_gemm_strided(SgArgs g) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int sp = blockIdx.y;
  if (e >= (int64_t)g.I * g.J) return;
  const int i = (int)(e / g.J), j = (int)(e % g.J);
  const int64_t c0 = (int64_t)sp * g.len_per_split, c1 = min(g.L, c0 + g.len_per_split);
  const float* a = g.A + (int64_t)i * g.sai;
  const float* b = g.B + (int64_t)j * g.sbj;
  float s = 0.f;
  for (int64_t c = c0; c < c1; ++c) s = fmaf(a[c * g.sac], b[c * g.sbc], s);
  if (g.splits > 1) g.C[((int64_t)sp * g.I + i) * g.J + j] = s;
  else g.C[(int64_t)i * g.ldc + j] = s;
}

static int sg_splits(int I, int J, int64_t L) {
  const int64_t outs = (int64_t)I * J;
  int64_t s = (256 * 256 * 4 + outs - 1) / outs;               // enough threads for the chip
  const int64_t most = (L + 63) / 64;
  if (s > most) s = most;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return (int)s;
}

extern "C" size_t sv_gemm_strided_scratch_bytes(int I, int J, int64_t L) {
  if (I <= 0 || J <= 0 || L <= 0) return 256;
  const int s = sg_splits(I, J, L);
  return dt_align((size_t)(s > 1 ? s : 0) * I * J * sizeof(float)) + 256;
}

extern "C" int sv_gemm_strided(const float* A, int64_t stride_a_row, int64_t stride_a_c, const float* B, int64_t stride_b_c, int64_t stride_b_col, float* C,
                               int64_t ldc, int rows, int cols, int64_t contraction, void* scratch, void* stream) {
  SV_CHECK_ARG(rows >= 0 && cols >= 0 && contraction >= 0 && ldc >= cols, "gemm_strided: bad sizes");
  if (rows == 0 || cols == 0) return SV_OK;
  SV_CHECK_ARG(C && scratch && (contraction == 0 || (A && B)), "gemm_strided: null pointer");
  hipStream_t st = sv_stream(stream);
  SgArgs g{A, stride_a_row, stride_a_c, B, stride_b_c, stride_b_col, C, ldc, rows, cols, sg_splits(rows, cols, contraction), contraction, 0};
  g.len_per_split = (contraction + g.splits - 1) / g.splits;
  if (g.splits > 1) g.C = reinterpret_cast<float*>(scratch);
  hipLaunchKernelGGL(k_gemm_strided, dim3(sv_div_up((int64_t)rows * cols, 256), g.splits), dim3(256), 0, st, g);
  if (g.splits > 1)
    hipLaunchKernelGGL(k_split_reduce, dim3(sv_grid_1d((int64_t)rows * cols, 256)), dim3(256), 0, st, reinterpret_cast<const float*>(scratch), g.splits, rows, cols, C, ldc);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// C (N, K) = A^T B,  A (M, N) rows lda apart, B (M, K) rows ldb apart.  The matrix-core path takes N % 4 == 0, K % 4 == 0, lda % 4 == 0, ldb % 4 == 0 and
// 16-byte aligned operands; anything else goes through sv_gemm_strided (same result up to summation order).
extern "C" int sv_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, void* scratch,
                          void* stream) {
  SV_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && lda >= N && ldb >= K && ldc >= K, "gemm_tn: bad sizes");
  if (N == 0 || K == 0) return SV_OK;
  SV_CHECK_ARG(C && scratch && (M == 0 || (A && B)), "gemm_tn: null pointer");
  hipStream_t st = sv_stream(stream);
  const bool mfma = M >= 64 && N % 4 == 0 && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) &&
                    (int64_t)N * K >= 64 * 64;
  if (!mfma) return sv_gemm_strided(A, 1, lda, B, ldb, 1, C, ldc, N, K, M, scratch, stream);
  TnArgs g{A, lda, B, ldb, C, ldc, M, N, K, tn_splits(M, N, K), 0};
  g.rows_per_split = ((M + g.splits - 1) / g.splits + TN_BM - 1) / TN_BM * TN_BM;
  g.splits = (int)((M + g.rows_per_split - 1) / g.rows_per_split);
  if (g.splits > 1) g.C = reinterpret_cast<float*>(scratch);
  hipLaunchKernelGGL(k_gemm_tn, dim3(sv_div_up(K, TN_T), sv_div_up(N, TN_T), g.splits), dim3(256), 0, st, g);
  if (g.splits > 1)
    hipLaunchKernelGGL(k_split_reduce, dim3(sv_grid_1d((int64_t)N * K, 256)), dim3(256), 0, st, reinterpret_cast<const float*>(scratch), g.splits, N, K, C, ldc);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------ column sums
// out[n] = sum_m A[m][n]: row chunks -> partial (chunks, N) -> fixed-order sum
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ A, int64_t lda, int64_t M, int N, int64_t rows_per_chunk, float* __restrict__ partial) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int64_t m0 = (int64_t)blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  float s = 0.f;
  for (int64_t m = m0; m < m1; ++m) s += A[m * lda + n];
  partial[(int64_t)blockIdx.y * N + n] = s;
}

static int colsum_chunks(int64_t M, int N) {
  int64_t c = (256 * 8 * 256) / ((int64_t)sv_div_up(N, 256) * 256);
  const int64_t most = (M + 63) / 64;
  if (c > most) c = most;
  if (c < 1) c = 1;
  if (c > 4096) c = 4096;
  return (int)c;
}

extern "C" size_t sv_column_sums_scratch_bytes(int64_t M, int N) { return dt_align((size_t)colsum_chunks(M < 1 ? 1 : M, N < 1 ? 1 : N) * (N < 1 ? 1 : N) * sizeof(float)) + 256; }

extern "C" int sv_column_sums(const float* A, int64_t lda, int64_t M, int N, float* out, void* scratch, void* stream) {
  SV_CHECK_ARG(M >= 0 && N >= 0 && lda >= N, "column_sums: bad sizes");
  if (N == 0) return SV_OK;
  SV_CHECK_ARG(out && scratch && (M == 0 || A), "column_sums: null pointer");
  hipStream_t st = sv_stream(stream);
  const int chunks = colsum_chunks(M < 1 ? 1 : M, N);
  const int64_t rpc = (M + chunks - 1) / chunks;
  float* partial = reinterpret_cast<float*>(scratch);
  hipLaunchKernelGGL(k_colsum_partial, dim3(sv_div_up(N, 256), chunks), dim3(256), 0, st, A, lda, M, N, rpc > 0 ? rpc : 1, partial);
  hipLaunchKernelGGL(k_split_reduce, dim3(sv_grid_1d(N, 256)), dim3(256), 0, st, partial, chunks, 1, N, out, (int64_t)N);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------ segments
// rows g * rows_per_group .. (g + 1) * rows_per_group - 1 form group g (an object's n points).  Max with the FIRST arg-max row (torch.max(dim) on
// CUDA returns an index of a maximal element; which one on ties is unspecified -- the gradient goes to exactly one of them either way).
__global__ __launch_bounds__(256) void k_segment_max(const float* __restrict__ x, int64_t ldx, int rows_per_group, int C, float* __restrict__ out,
                                                     int32_t* __restrict__ arg) {
  const int c = blockIdx.x * 256 + threadIdx.x, grp = blockIdx.y;
  if (c >= C) return;
  const float* p = x + (int64_t)grp * rows_per_group * ldx + c;
  float best = p[0];
  int bi = 0;
  for (int r = 1; r < rows_per_group; ++r) {
    const float v = p[(int64_t)r * ldx];
    if (v > best || (v != v && best == best)) best = v, bi = r;          // NaN propagates like torch.max
  }
  out[(int64_t)grp * C + c] = best;
  arg[(int64_t)grp * C + c] = bi;
}

extern "C" int sv_segment_max(const float* x, int64_t ldx, int groups, int rows_per_group, int channels, float* out, int32_t* arg, void* stream) {
  SV_CHECK_ARG(groups >= 0 && rows_per_group >= 1 && channels >= 0 && ldx >= channels, "segment_max: bad sizes");
  if (groups == 0 || channels == 0) return SV_OK;
  SV_CHECK_ARG(x && out && arg, "segment_max: null pointer");
  hipLaunchKernelGGL(k_segment_max, dim3(sv_div_up(channels, 256), groups), dim3(256), 0, sv_stream(stream), x, ldx, rows_per_group, channels, out, arg);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// dx (groups * rows_per_group, C) = 0 except dx[g * rpg + arg[g][c]][c] = dout[g][c]: every element written exactly once
__global__ __launch_bounds__(256) void k_segment_max_bwd(const float* __restrict__ dout, const int32_t* __restrict__ arg, int rows_per_group, int C,
                                                         float* __restrict__ dx, int64_t ldx) {
  const int c = blockIdx.x * 256 + threadIdx.x, grp = blockIdx.y;
  if (c >= C) return;
  const int a = arg[(int64_t)grp * C + c];
  const float gv = dout[(int64_t)grp * C + c];
  float* p = dx + (int64_t)grp * rows_per_group * ldx + c;
  for (int r = 0; r < rows_per_group; ++r) p[(int64_t)r * ldx] = r == a ? gv : 0.f;
}

extern "C" int sv_segment_max_backward(const float* dout, const int32_t* arg, int groups, int rows_per_group, int channels, float* dx, int64_t ldx,
                                       void* stream) {
  SV_CHECK_ARG(groups >= 0 && rows_per_group >= 1 && channels >= 0 && ldx >= channels, "segment_max_backward: bad sizes");
  if (groups == 0 || channels == 0) return SV_OK;
  SV_CHECK_ARG(dout && arg && dx, "segment_max_backward: null pointer");
  hipLaunchKernelGGL(k_segment_max_bwd, dim3(sv_div_up(channels, 256), groups), dim3(256), 0, sv_stream(stream), dout, arg, rows_per_group, channels, dx, ldx);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// out[g][c] = sum over the group's rows of x[row][c], rows in order (reproducible)
__global__ __launch_bounds__(256) void k_segment_sum(const float* __restrict__ x, int64_t ldx, int rows_per_group, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x, grp = blockIdx.y;
  if (c >= C) return;
  const float* p = x + (int64_t)grp * rows_per_group * ldx + c;
  float s = 0.f;
  for (int r = 0; r < rows_per_group; ++r) s += p[(int64_t)r * ldx];
  out[(int64_t)grp * C + c] = s;
}

extern "C" int sv_segment_sum(const float* x, int64_t ldx, int groups, int rows_per_group, int channels, float* out, void* stream) {
  SV_CHECK_ARG(groups >= 0 && rows_per_group >= 1 && channels >= 0 && ldx >= channels, "segment_sum: bad sizes");
  if (groups == 0 || channels == 0) return SV_OK;
  SV_CHECK_ARG(x && out, "segment_sum: null pointer");
  hipLaunchKernelGGL(k_segment_sum, dim3(sv_div_up(channels, 256), groups), dim3(256), 0, sv_stream(stream), x, ldx, rows_per_group, channels, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// dz = dy * act'(y): ReLU / LeakyReLU derivative taken from the layer's OUTPUT (y > 0 <=> pre-activation > 0 for both); act 0 copies
__global__ __launch_bounds__(256) void k_act_backward(const float* __restrict__ dy, const float* __restrict__ y, int64_t n, int act, float slope, float* __restrict__ dz) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float g = dy[i], v = y[i];
    dz[i] = act == SV_ACT_RELU ? (v > 0.f ? g : 0.f) : act == SV_ACT_LRELU ? (v > 0.f ? g : g * slope) : g;     // torch: input > 0 takes the unit slope, 0 and -0 take `slope`
  }
}

extern "C" int sv_act_backward(const float* dy, const float* y, int64_t n, int act, float slope, float* dz, void* stream) {
  SV_CHECK_ARG(n >= 0 && act >= 0 && act <= 2, "act_backward: bad arguments");
  if (n == 0) return SV_OK;
  SV_CHECK_ARG(dy && y && dz, "act_backward: null pointer");
  hipLaunchKernelGGL(k_act_backward, dim3(sv_grid_1d(n, 256)), dim3(256), 0, sv_stream(stream), dy, y, n, act, slope, dz);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
