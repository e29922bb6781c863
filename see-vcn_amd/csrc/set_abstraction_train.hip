// Set abstraction, TRAINING mode: one radius scale of StackSAModuleMSG.forward and its backward without the (M, C+3, nsample) tensor and
// without a library GEMM.  Replaces (detector3d/pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:78-112)
//   QueryAndGroup (pointnet2_utils.py:112-159; group_points_kernel_stack, src/group_points_gpu.cu:71-102: the gathered (M, C+3, ns) tensor)
//   -> Conv2d 1x1 -> BatchNorm2d (BATCH statistics over all M * ns positions, empty balls = zero rows included) -> ReLU, twice
//   -> max_pool2d over nsample,
// and the autograd chain behind it (cuDNN convs + batch_norm_backward + group_points_grad_kernel_stack's atomicAdd scatter).
//
// BatchNorm in training mode needs every row's pre-activation before any row can be normalised, so a scale is a chain of passes over
// 16-row tiles (a query's nsample neighbours = 1 or 2 tiles), each pass one persistent launch whose waves keep their accumulators in registers:
//   proj   layer 1 is linear in the gathered row, and the feature part of a gathered row is a COPY of a support point's row: P = F . W1[:, 3:]^T
//          is made once per support point (N x C x C1 on the matrix core instead of R x C x C1, R = M * nsample = 27 .. 108 N here)
//   fwd1   z1[row] = P[source of row] + W1[:, :3] . (xyz[source] - query)  (R, C1) + per-channel sum / sum of squares: a gather, no GEMM
//   (fin)  k_bn_finalize (norm.hip): batch mean / invstd, running statistics, scale / shift
//   fwd2   z1 -> scale, shift, ReLU fused into the operand load -> layer-2 GEMM -> z2 (R, C2) + statistics + per query and channel the largest
//          and the smallest z2 with their slots: max over slots of relu(s * z + t) = relu(s * (s >= 0 ? max z : min z) + t), bit for bit
//   (fin)  out   out (M, C2), selected slot, selected z2
//   bwd0   the gradient enters at ONE slot per (query, channel): BatchNorm-2 backward sums over (M, C2) only
//   bwd2   tiles: dz2 (BatchNorm-2 backward, dense) -> weight gradient 2 (contraction over rows on the matrix core, accumulators stay in
//          registers for the whole launch) and da1 = dz2 . W2 -> ReLU mask -> dy1 (R, C1) + BatchNorm-1 backward sums
//   bwd1   tiles: dz1 -> S[source row] += dz1 (float atomics, one 256-byte row per instruction; the reference scatters with atomicAdd too:
//          group_points_gpu.cu:38-41) and the xyz columns of the weight gradient (sum of dz1 x relative coordinate).  Everything else of
//          layer 1's backward is linear in the gathered row and happens per SUPPORT POINT afterwards: dW1[:, 3:] = S^T . F (wgp) and
//          dF = S . W1[:, 3:] (fgrad) are small GEMMs over the N points instead of passes over (R, C) tensors.
// Only C1- / C2-wide per-row tensors touch HBM (z1, z2, dy1: 256 B per row at 64 channels); the (C+3)-wide gathered rows never do.
// At the RoI-grid pool (pvrcnn_head.py:64-109: 110 592 queries x 16 neighbours, C = 128, 64/64 channels): 43.5 GFLOP per scale on the fp32
// matrix core (layer 2 forward, its data and weight gradients) instead of the 109 GFLOP of the literal formulation, 9 x 453 MB of per-row
// traffic (z1, z2, dy1).
#include "norm.h"

typedef float st_f4 __attribute__((ext_vector_type(4)));

constexpr int ST_THREADS = 512;              // 8 waves: two per SIMD, one workgroup per CU
constexpr int ST_WAVES = ST_THREADS / 64;
constexpr int ST_MAXC = 64;                  // widest MLP layer
constexpr int ST_NT = ST_MAXC / 16;
constexpr int ST_MAXF = 128;                 // most feature channels

constexpr int ST_PITCH = ST_MAXC + 4;        // LDS row pitch of a 16-row tile
constexpr int ST_GRID = 512;                 // two workgroups per CU: four waves per SIMD (kernels hold <= 128 VGPRs, <= 80 KB LDS)

struct SaT {
  const float* xyz;          // (N, 3)
  const float* feat;         // (N, C) or null
  const float* new_xyz;      // (M, 3)
  const int32_t* idx;        // (M, ns) scene-local, idx[q][0] < 0: empty ball
  const int32_t* row_start;  // (M)
  const float* w1;           // (C1, 3 + C) parameter layout (xyz columns first)
  const float* w2;           // (C2, C1)
  float* z1;                 // (R, C1)
  float* z2;                 // (R, C2)
  float* dy1;                // (R, C1)
  float* part;               // statistics partials of this pass (grid, 2, Cx)
  const float* coef1;        // fwd: {scale, shift} of BatchNorm 1; bwd: see bwd kernels
  const float* coef2;
  const float* istd1;        // save_invstd / save_mean of BatchNorm 1, 2
  const float* mean1;
  const float* istd2;
  const float* mean2;
  float* zmax;               // (M, C2)  (after k_sa_out: the selected z2)
  float* zmin;               // (M, C2)  (after k_sa_bwd0: the masked output gradient)
  uint8_t* amax;             // (M, C2)  (after k_sa_out: the selected slot)
  uint8_t* amin;
  float* out;                // (M, C2)
  const float* dout;         // (M, C2)
  float* wpart;              // weight-gradient partials
  float* S;                  // (N, C1) scatter target of bwd1
  float* P;                  // (N, C1) layer-1 feature part per support point (forward)
  int64_t M, N;
  int C, Kp, C1, C2, ns;
};

__device__ __forceinline__ st_f4 mfma4(const st_f4 a, const st_f4 b, st_f4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  return c;
}

// 16 x C tile: LDS (pitch ST_PITCH) -> its contiguous place in a (R, C) matrix, whole 16-byte pieces in address order
__device__ __forceinline__ void tile_to_global(const float* T, float* dst, int C, int lane) {
  const int c4n = C >> 2, nf = 16 * c4n;
  for (int f = lane; f < nf; f += 64) {
    const int row = f / c4n, c4 = f - row * c4n;
    reinterpret_cast<st_f4*>(dst)[f] = *reinterpret_cast<const st_f4*>(T + row * ST_PITCH + c4 * 4);
  }
}

// accumulator tiles (D layout: rows 4 kk + r, column 16 t + li) -> LDS tile
__device__ __forceinline__ void acc_to_lds(float* T, const st_f4 (&acc)[ST_NT], int nt, int li, int kk) {
#pragma unroll
  for (int t = 0; t < ST_NT; ++t)
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(kk * 4 + r) * ST_PITCH + t * 16 + li] = acc[t][r];
    }
}

// per-lane column sums -> workgroup partial (grid, 2, C): across the 4 row groups of a wave by shuffles, across waves in a fixed order
__device__ __forceinline__ void stats_to_partial(float (&s0)[ST_NT], float (&s1)[ST_NT], int nt, int C, float (*s_red)[2][ST_MAXC], float* part, int wid,
                                                 int lane) {
  const int li = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int t = 0; t < ST_NT; ++t)
    if (t < nt) {
      float a = s0[t], b = s1[t];
      a += __shfl_xor(a, 16), b += __shfl_xor(b, 16);
      a += __shfl_xor(a, 32), b += __shfl_xor(b, 32);
      if (kk == 0) s_red[wid][0][t * 16 + li] = a, s_red[wid][1][t * 16 + li] = b;
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * C; e += ST_THREADS) {
    const int which = e / C, c = e - which * C;
    float v = 0.f;
    for (int w = 0; w < ST_WAVES; ++w) v += s_red[w][which][c];
    part[(size_t)blockIdx.x * 2 * C + which * C + c] = v;
  }
}

// ------------------------------------------------------------------------------------------------ forward, layer 1
// P (N, C1) = F . W1[:, 3:]^T on the matrix core: 16 support points per wave tile
constexpr int SP_THREADS = 256;
__global__ __launch_bounds__(SP_THREADS) void k_sa_point_proj(SaT a) {
  __shared__ __attribute__((aligned(16))) float s_w[ST_MAXC * (ST_MAXF + 4)];             // (C1, C + 4): feature columns of W1
  __shared__ __attribute__((aligned(16))) float s_t[SP_THREADS / 64][16 * ST_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, kk = lane >> 4;
  const int pw = a.C + 4, cin = a.C + 3;
  for (int e = tid; e < a.C1 * a.C; e += SP_THREADS) s_w[(e / a.C) * pw + e % a.C] = a.w1[(e / a.C) * cin + 3 + e % a.C];
  __syncthreads();
  float* T = s_t[wid];
  const int nt1 = a.C1 / 16, nqf = a.C / 16;
  const int64_t ntiles = (a.N + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * (SP_THREADS / 64) + wid; tile < ntiles; tile += (int64_t)gridDim.x * (SP_THREADS / 64)) {
    const int64_t n = tile * 16 + li;
    st_f4 acc[ST_NT];
#pragma unroll
    for (int t = 0; t < ST_NT; ++t) acc[t] = (st_f4){0.f, 0.f, 0.f, 0.f};
    for (int qs = 0; qs < nqf; ++qs) {
      st_f4 A = (st_f4){0.f, 0.f, 0.f, 0.f};
      if (n < a.N) A = *reinterpret_cast<const st_f4*>(a.feat + n * a.C + qs * 16 + kk * 4);
#pragma unroll
      for (int t = 0; t < ST_NT; ++t)
        if (t < nt1) acc[t] = mfma4(A, *reinterpret_cast<const st_f4*>(s_w + (t * 16 + li) * pw + qs * 16 + kk * 4), acc[t]);
    }
    acc_to_lds(T, acc, nt1, li, kk);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const int rows = (int)min((int64_t)16, a.N - tile * 16);           // the last tile may be short
    const int c4n = a.C1 >> 2;
    for (int f = lane; f < rows * c4n; f += 64) {
      const int row = f / c4n, c4 = f - row * c4n;
      reinterpret_cast<st_f4*>(a.P + tile * 16 * a.C1)[f] = *reinterpret_cast<const st_f4*>(T + row * ST_PITCH + c4 * 4);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
}

// z1[row] = P[source] + W1[:, :3] . (xyz[source] - query), 0 for an empty ball; per-channel sum / sum of squares.  Lane (row li, columns
// 16 qs + 4 kk .. + 3): 16-byte pieces of the P rows in, 16-byte pieces of z1 out; no matrix work.
constexpr int SG_THREADS = 256;
template <int NT1, int NT2>   // channel tiles of the two layers at compile time (0: read from the arguments)
__global__ __launch_bounds__(SG_THREADS) void k_sa_fwd1(SaT a_in) {
  SaT a = a_in;
  if (NT1) a.C1 = NT1 * 16;
  if (NT2) a.C2 = NT2 * 16;
  __shared__ __attribute__((aligned(16))) float s_wx[3][ST_MAXC];                        // xyz columns of W1, by coordinate
  __shared__ float s_red[SG_THREADS / 64][2][ST_MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, kk = lane >> 4;
  for (int e = tid; e < 3 * ST_MAXC; e += SG_THREADS) s_wx[e / ST_MAXC][e % ST_MAXC] = (e % ST_MAXC) < a.C1 ? a.w1[(e % ST_MAXC) * (a.C + 3) + e / ST_MAXC] : 0.f;
  __syncthreads();
  const int nt1 = NT1 ? NT1 : a.C1 / 16, G = a.ns / 16;
  st_f4 s0[ST_NT], s1[ST_NT];
#pragma unroll
  for (int qs = 0; qs < ST_NT; ++qs) s0[qs] = s1[qs] = (st_f4){0.f, 0.f, 0.f, 0.f};
  for (int64_t q = (int64_t)blockIdx.x * (SG_THREADS / 64) + wid; q < a.M; q += (int64_t)gridDim.x * (SG_THREADS / 64)) {
    const bool empty = a.idx[q * a.ns] < 0;
    const int64_t base = a.row_start[q];
    const float qx = a.new_xyz[q * 3], qy = a.new_xyz[q * 3 + 1], qz = a.new_xyz[q * 3 + 2];
    for (int g = 0; g < G; ++g) {
      const int64_t row = q * a.ns + g * 16 + li;
      float dx = 0.f, dy = 0.f, dz = 0.f;
      int64_t src = -1;
      if (!empty) {
        src = base + a.idx[row];
        const float* p = a.xyz + src * 3;
        dx = p[0] - qx, dy = p[1] - qy, dz = p[2] - qz;
      }
#pragma unroll
      for (int qs = 0; qs < ST_NT; ++qs)
        if (qs < nt1) {
          const int c = qs * 16 + kk * 4;
          st_f4 z = (st_f4){0.f, 0.f, 0.f, 0.f};
          if (!empty) {
            if (a.P) z = *reinterpret_cast<const st_f4*>(a.P + src * a.C1 + c);
            const st_f4 wx = *reinterpret_cast<const st_f4*>(s_wx[0] + c), wy = *reinterpret_cast<const st_f4*>(s_wx[1] + c),
                        wz = *reinterpret_cast<const st_f4*>(s_wx[2] + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = fmaf(wz[i], dz, fmaf(wy[i], dy, fmaf(wx[i], dx, z[i])));
          }
          *reinterpret_cast<st_f4*>(a.z1 + row * a.C1 + c) = z;
#pragma unroll
          for (int i = 0; i < 4; ++i) s0[qs][i] += z[i], s1[qs][i] += z[i] * z[i];
        }
    }
  }
  // column sums: over the 16 rows of a wave by shuffles, over the waves in a fixed order
#pragma unroll
  for (int qs = 0; qs < ST_NT; ++qs)
    if (qs < nt1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float u = s0[qs][i], v = s1[qs][i];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) u += __shfl_xor(u, off), v += __shfl_xor(v, off);
        if (li == 0) s_red[wid][0][qs * 16 + kk * 4 + i] = u, s_red[wid][1][qs * 16 + kk * 4 + i] = v;
      }
    }
  __syncthreads();
  for (int e = tid; e < 2 * a.C1; e += SG_THREADS) {
    const int which = e / a.C1, c = e - which * a.C1;
    float v = 0.f;
    for (int w = 0; w < SG_THREADS / 64; ++w) v += s_red[w][which][c];
    a.part[(size_t)blockIdx.x * 2 * a.C1 + which * a.C1 + c] = v;
  }
}

// ------------------------------------------------------------------------------------------------ forward, layer 2
template <int NT1, int NT2>   // channel tiles of the two layers at compile time (0: read from the arguments)
__global__ __launch_bounds__(ST_THREADS, 4) void k_sa_fwd2(SaT a_in) {
  SaT a = a_in;
  if (NT1) a.C1 = NT1 * 16;
  if (NT2) a.C2 = NT2 * 16;
  __shared__ __attribute__((aligned(16))) float s_w2[ST_MAXC * (ST_MAXC + 4)];            // (C2, p2)
  __shared__ __attribute__((aligned(16))) float s_t[ST_WAVES][16 * ST_PITCH];
  __shared__ float s_red[ST_WAVES][2][ST_MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, kk = lane >> 4;
  __shared__ __attribute__((aligned(16))) float s_c1[2][ST_MAXC];                         // BatchNorm-1 scale / shift (read per operand load: registers are short)
  const int p2 = a.C1 + 4;
  for (int e = tid; e < a.C2 * a.C1; e += ST_THREADS) s_w2[(e / a.C1) * p2 + e % a.C1] = a.w2[e];
  for (int e = tid; e < a.C1; e += ST_THREADS) s_c1[0][e] = a.coef1[e], s_c1[1][e] = a.coef1[a.C1 + e];
  __syncthreads();
  float* T = s_t[wid];
  const int nt1 = NT1 ? NT1 : a.C1 / 16, nt2 = NT2 ? NT2 : a.C2 / 16, G = a.ns / 16;
  float s0[ST_NT] = {0.f, 0.f, 0.f, 0.f}, s1[ST_NT] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t q = (int64_t)blockIdx.x * ST_WAVES + wid; q < a.M; q += (int64_t)gridDim.x * ST_WAVES) {
    float vmax[ST_NT], vmin[ST_NT];
    int smax[ST_NT], smin[ST_NT];
#pragma unroll
    for (int t = 0; t < ST_NT; ++t) vmax[t] = -3.4e38f, vmin[t] = 3.4e38f, smax[t] = smin[t] = 0;
    for (int g = 0; g < G; ++g) {
      const int64_t rowbase = q * a.ns + g * 16;
      st_f4 acc[ST_NT];
#pragma unroll
      for (int t = 0; t < ST_NT; ++t) acc[t] = (st_f4){0.f, 0.f, 0.f, 0.f};
      st_f4 Az[ST_NT];                                               // the tile's z1 rows, all requested before the first use
#pragma unroll
      for (int qs = 0; qs < ST_NT; ++qs) {
        Az[qs] = (st_f4){0.f, 0.f, 0.f, 0.f};
        if (qs < nt1) Az[qs] = *reinterpret_cast<const st_f4*>(a.z1 + (rowbase + li) * a.C1 + qs * 16 + kk * 4);
      }
#pragma unroll
      for (int qs = 0; qs < ST_NT; ++qs)
        if (qs < nt1) {
          const st_f4 sc = *reinterpret_cast<const st_f4*>(s_c1[0] + qs * 16 + kk * 4), sh = *reinterpret_cast<const st_f4*>(s_c1[1] + qs * 16 + kk * 4);
          st_f4 A = Az[qs];
          A.x = fmaxf(fmaf(A.x, sc.x, sh.x), 0.f), A.y = fmaxf(fmaf(A.y, sc.y, sh.y), 0.f);
          A.z = fmaxf(fmaf(A.z, sc.z, sh.z), 0.f), A.w = fmaxf(fmaf(A.w, sc.w, sh.w), 0.f);
#pragma unroll
          for (int t = 0; t < ST_NT; ++t)
            if (t < nt2) acc[t] = mfma4(A, *reinterpret_cast<const st_f4*>(s_w2 + (t * 16 + li) * p2 + qs * 16 + kk * 4), acc[t]);
        }
#pragma unroll
      for (int t = 0; t < ST_NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[t][r];
          const int slot = g * 16 + kk * 4 + r;
          s0[t] += v, s1[t] += v * v;
          if (v > vmax[t]) vmax[t] = v, smax[t] = slot;       // slots ascend inside a lane: strict comparisons keep the first occurrence
          if (v < vmin[t]) vmin[t] = v, smin[t] = slot;
        }
      acc_to_lds(T, acc, nt2, li, kk);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      tile_to_global(T, a.z2 + rowbase * a.C2, a.C2, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
#pragma unroll
    for (int t = 0; t < ST_NT; ++t)
      if (t < nt2) {
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
          const float ov = __shfl_xor(vmax[t], off), ow = __shfl_xor(vmin[t], off);
          const int os = __shfl_xor(smax[t], off), ot = __shfl_xor(smin[t], off);
          if (ov > vmax[t] || (ov == vmax[t] && os < smax[t])) vmax[t] = ov, smax[t] = os;
          if (ow < vmin[t] || (ow == vmin[t] && ot < smin[t])) vmin[t] = ow, smin[t] = ot;
        }
        if (kk == 0) {
          const int64_t o = q * a.C2 + t * 16 + li;
          a.zmax[o] = vmax[t], a.zmin[o] = vmin[t], a.amax[o] = (uint8_t)smax[t], a.amin[o] = (uint8_t)smin[t];
        }
      }
  }
  stats_to_partial(s0, s1, nt2, a.C2, s_red, a.part, wid, lane);
}

// out = relu(scale * sel + shift), sel = the largest z2 of the query where scale >= 0, the smallest where it is negative
__global__ __launch_bounds__(256) void k_sa_out(SaT a) {
  const int64_t total = a.M * a.C2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % a.C2);
    const float sc = a.coef2[c], sh = a.coef2[a.C2 + c];
    const bool pos = sc >= 0.f;
    const float sel = pos ? a.zmax[i] : a.zmin[i];
    a.out[i] = fmaxf(fmaf(sel, sc, sh), 0.f);
    a.zmax[i] = sel;
    a.amax[i] = pos ? a.amax[i] : a.amin[i];
  }
}

// ------------------------------------------------------------------------------------------------ backward
// bwd0: the output gradient reaches z2 at one slot per (query, channel), where the output is positive: masked gradient -> zmin (reused),
// BatchNorm-2 backward sums (sum dy, sum dy * xhat) over the (M, C2) entries -> partial (grid, 2, C2)
__global__ __launch_bounds__(256) void k_sa_bwd0(SaT a) {
  __shared__ float s_r[2][256];
  const int tid = threadIdx.x, C = a.C2;
  const int rpw = 256 / C;                                          // C in {16, 32, 48, 64}: whole rows per sweep, idle tail lanes for 48
  const int c = tid % C, rr = tid / C;
  const bool active = rr < rpw;
  const int64_t rows_per_wg = (a.M + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = min(a.M, r0 + rows_per_wg);
  float t0 = 0.f, t1 = 0.f;
  if (active) {
    const float m = a.mean2[c], is = a.istd2[c];
    for (int64_t r = r0 + rr; r < r1; r += rpw) {
      const int64_t i = r * C + c;
      const float d = a.out[i] > 0.f ? a.dout[i] : 0.f;
      a.zmin[i] = d;
      t0 += d, t1 += d * ((a.zmax[i] - m) * is);
    }
  }
  s_r[0][tid] = active ? t0 : 0.f, s_r[1][tid] = active ? t1 : 0.f;
  __syncthreads();
  if (tid < C) {
    float u0 = 0.f, u1 = 0.f;
    for (int k = 0; k < rpw; ++k) u0 += s_r[0][k * C + tid], u1 += s_r[1][k * C + tid];
    a.part[(size_t)blockIdx.x * 2 * C + tid] = u0;
    a.part[(size_t)blockIdx.x * 2 * C + C + tid] = u1;
  }
}

// bwd2.  coef2 = k_bn_finalize<true>'s {gamma*invstd, mean(dy), mean(dy*xhat), mean} of BatchNorm 2; coef1 = {scale, shift} of BatchNorm 1.
// dz2 = k * (dy - md - xhat * mx) = k * dy + A * z + B  with  A = -k * mx * invstd,  B = k * (mx * invstd * mean - md)
// Waves w and w + 4 of a workgroup take the same queries and split the layer-1 channels (the columns of da1 / dy1 / dW2) between them: each
// makes the cheap dz2 tile for itself and runs half of the matrix work, with its 32 + 16 accumulator registers instead of 64 + 32 --
// four waves per SIMD instead of two with spills.
constexpr int ST_NTH = ST_NT / 2;
template <int NT1, int NT2>   // channel tiles of the two layers at compile time (0: read from the arguments)
__global__ __launch_bounds__(ST_THREADS, 4) void k_sa_bwd2(SaT a_in) {
  SaT a = a_in;
  if (NT1) a.C1 = NT1 * 16;
  if (NT2) a.C2 = NT2 * 16;
  __shared__ __attribute__((aligned(16))) float s_w2t[ST_MAXC * (ST_MAXC + 4)];           // (C1, pT): W2 transposed
  __shared__ __attribute__((aligned(16))) float s_t[ST_WAVES][16 * ST_PITCH];
  __shared__ float s_red[ST_WAVES][2][ST_MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, kk = lane >> 4;
  const int pair = wid & 3, half = wid >> 2;
  const int pT = a.C2 + 4;
  for (int e = tid; e < a.C2 * a.C1; e += ST_THREADS) s_w2t[(e % a.C1) * pT + e / a.C1] = a.w2[e];
  __syncthreads();
  float* T = s_t[wid];
  const int nt1 = NT1 ? NT1 : a.C1 / 16, nt2 = NT2 ? NT2 : a.C2 / 16, G = a.ns / 16;
  const int u0 = half ? (nt1 + 1) / 2 : 0, u1 = half ? nt1 : (nt1 + 1) / 2;              // this wave's layer-1 channel tiles
  // per-channel constants live in LDS (seven per channel would cost 28 registers a lane): s_k[0..2] = k2, A2, B2; s_k[3..6] = scale1, shift1,
  // invstd1, -mean1 * invstd1
  __shared__ float s_k[7][ST_MAXC];
  for (int c = tid; c < ST_MAXC; c += ST_THREADS) {
    float k = 0.f, A = 0.f, B = 0.f, s1v = 0.f, h1 = 0.f, i1 = 0.f, n1 = 0.f;
    if (c < a.C2) {
      const float md = a.coef2[a.C2 + c], mx = a.coef2[2 * a.C2 + c], m = a.coef2[3 * a.C2 + c], is = a.istd2[c];
      k = a.coef2[c], A = -k * mx * is, B = k * (mx * is * m - md);
    }
    if (c < a.C1) s1v = a.coef1[c], h1 = a.coef1[a.C1 + c], i1 = a.istd1[c], n1 = -a.mean1[c] * a.istd1[c];
    s_k[0][c] = k, s_k[1][c] = A, s_k[2][c] = B, s_k[3][c] = s1v, s_k[4][c] = h1, s_k[5][c] = i1, s_k[6][c] = n1;
  }
  __syncthreads();
  st_f4 accw[ST_NT][ST_NTH];                                         // dW2[c2 = 16 t + 4 kk + r][c1 = 16 (u0 + j) + li]
#pragma unroll
  for (int t = 0; t < ST_NT; ++t)
#pragma unroll
    for (int j = 0; j < ST_NTH; ++j) accw[t][j] = (st_f4){0.f, 0.f, 0.f, 0.f};
  float sd[ST_NTH] = {0.f, 0.f}, sdx[ST_NTH] = {0.f, 0.f};
  for (int64_t q = (int64_t)blockIdx.x * 4 + pair; q < a.M; q += (int64_t)gridDim.x * 4) {
    float dq[ST_NT];
    int aq[ST_NT];
#pragma unroll
    for (int t = 0; t < ST_NT; ++t) {
      dq[t] = 0.f, aq[t] = -1;
      if (t < nt2) dq[t] = a.zmin[q * a.C2 + t * 16 + li], aq[t] = a.amax[q * a.C2 + t * 16 + li];
    }
    for (int g = 0; g < G; ++g) {
      const int64_t rowbase = q * a.ns + g * 16;
      st_f4 dz2[ST_NT];                                              // D layout: row 4 kk + r, column 16 t + li
#pragma unroll
      for (int t = 0; t < ST_NT; ++t) {
        dz2[t] = (st_f4){0.f, 0.f, 0.f, 0.f};
        if (t < nt2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = a.z2[(rowbase + kk * 4 + r) * a.C2 + t * 16 + li];
            const float dy = aq[t] == g * 16 + kk * 4 + r ? dq[t] : 0.f;
            dz2[t][r] = fmaf(s_k[0][t * 16 + li], dy, fmaf(s_k[1][t * 16 + li], z, s_k[2][t * 16 + li]));
          }
        }
      }
      st_f4 z1v[ST_NTH];                                             // this wave's z1 columns, requested before the matrix work
#pragma unroll
      for (int j = 0; j < ST_NTH; ++j) {
        z1v[j] = (st_f4){0.f, 0.f, 0.f, 0.f};
        if (u0 + j < u1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) z1v[j][r] = a.z1[(rowbase + kk * 4 + r) * a.C1 + (u0 + j) * 16 + li];
        }
      }
      acc_to_lds(T, dz2, nt2, li, kk);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      st_f4 da1[ST_NTH];                                             // da1 = dz2 . W2: rows x this wave's C1 columns
#pragma unroll
      for (int j = 0; j < ST_NTH; ++j) da1[j] = (st_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int qs = 0; qs < ST_NT; ++qs)
        if (qs < nt2) {
          const st_f4 A = *reinterpret_cast<const st_f4*>(T + li * ST_PITCH + qs * 16 + kk * 4);
#pragma unroll
          for (int j = 0; j < ST_NTH; ++j)
            if (u0 + j < u1) da1[j] = mfma4(A, *reinterpret_cast<const st_f4*>(s_w2t + ((u0 + j) * 16 + li) * pT + qs * 16 + kk * 4), da1[j]);
        }
      st_f4 a1[ST_NTH];
#pragma unroll
      for (int j = 0; j < ST_NTH; ++j) {
        a1[j] = (st_f4){0.f, 0.f, 0.f, 0.f};
        if (u0 + j < u1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = (u0 + j) * 16 + li;
            const float z = z1v[j][r];
            const float y = fmaf(z, s_k[3][c], s_k[4][c]);
            a1[j][r] = fmaxf(y, 0.f);
            const float d = y > 0.f ? da1[j][r] : 0.f;
            da1[j][r] = d;
            sd[j] += d, sdx[j] += d * fmaf(z, s_k[5][c], s_k[6][c]);
          }
        }
      }
      // weight gradient 2: dW2 += dz2^T . a1, contraction over the tile's rows (row 4 kk + s is the k index of MFMA s)
#pragma unroll
      for (int t = 0; t < ST_NT; ++t)
        if (t < nt2) {
#pragma unroll
          for (int j = 0; j < ST_NTH; ++j)
            if (u0 + j < u1) accw[t][j] = mfma4(dz2[t], a1[j], accw[t][j]);
        }
      // dy1: this wave's columns of the tile -> LDS (rows 4 kk + r) -> whole 128-byte pieces of the (R, C1) matrix
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // the A-operand reads of T are done (in order) before it is rewritten
#pragma unroll
      for (int j = 0; j < ST_NTH; ++j)
        if (u0 + j < u1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) T[(kk * 4 + r) * ST_PITCH + j * 16 + li] = da1[j][r];
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      const int c4n = (u1 - u0) * 4, nf = 16 * c4n;                    // float4 pieces per row of this wave's share
      for (int f = lane; f < nf; f += 64) {
        const int row = f / c4n, c4 = f - row * c4n;
        *reinterpret_cast<st_f4*>(a.dy1 + (rowbase + row) * a.C1 + u0 * 16 + c4 * 4) = *reinterpret_cast<const st_f4*>(T + row * ST_PITCH + c4 * 4);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
  }
  // weight-gradient partial of the workgroup (grid, C2, C1): 16 gradient rows at a time through the waves' LDS tiles; a column's sum runs over the
  // four waves of the half that owns it, in a fixed order
  float* wp = a.wpart + (size_t)blockIdx.x * a.C2 * a.C1;
  const int split = ((nt1 + 1) / 2) * 16;                              // first column of the second half
#pragma unroll
  for (int t = 0; t < ST_NT; ++t)
    if (t < nt2) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < ST_NTH; ++j)
        if (u0 + j < u1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) T[(kk * 4 + r) * ST_PITCH + (u0 + j) * 16 + li] = accw[t][j][r];
        }
      __syncthreads();
      for (int e = tid; e < 16 * a.C1; e += ST_THREADS) {
        const int row = e / a.C1, col = e - row * a.C1;
        const int w0 = col < split ? 0 : 4;
        float v = 0.f;
        for (int w = w0; w < w0 + 4; ++w) v += s_t[w][row * ST_PITCH + col];
        wp[(t * 16 + row) * a.C1 + col] = v;
      }
    }
  __syncthreads();
  // BatchNorm-1 backward sums: a wave holds its own channel tiles only
  float f0[ST_NT] = {0.f, 0.f, 0.f, 0.f}, f1[ST_NT] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < ST_NTH; ++j)
#pragma unroll
    for (int t = 0; t < ST_NT; ++t)
      if (t == u0 + j && u0 + j < u1) f0[t] = sd[j], f1[t] = sdx[j];
  stats_to_partial(f0, f1, nt1, a.C1, s_red, a.part, wid, lane);
}

// bwd1.  coef1 = k_bn_finalize<true>'s {gamma*invstd, mean(dy), mean(dy*xhat), mean} of BatchNorm 1: dz1 = k dy1 + A z1 + B per channel.
// Lane (row li, columns 16 qs + 4 kk .. + 3).  Per tile: dz1 -> LDS -> S[source of row] += dz1 row (one C1-wide row per atomic instruction, the
// repeats of the first neighbour summed in registers first), and wx[c][d] += dz1[row][c] * (xyz[source] - query)[d] -- the xyz columns of dW1.
template <int NT1, int NT2>   // channel tiles of the two layers at compile time (0: read from the arguments)
__global__ __launch_bounds__(SG_THREADS) void k_sa_bwd1(SaT a_in) {
  SaT a = a_in;
  if (NT1) a.C1 = NT1 * 16;
  if (NT2) a.C2 = NT2 * 16;
  __shared__ __attribute__((aligned(16))) float s_k[3][ST_MAXC];
  __shared__ __attribute__((aligned(16))) float s_t[SG_THREADS / 64][16 * ST_PITCH];
  __shared__ float s_wxr[SG_THREADS / 64][3][ST_MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, kk = lane >> 4;
  for (int c = tid; c < ST_MAXC; c += SG_THREADS) {
    float k = 0.f, A = 0.f, B = 0.f;
    if (c < a.C1) {
      const float md = a.coef1[a.C1 + c], mx = a.coef1[2 * a.C1 + c], m = a.coef1[3 * a.C1 + c], is = a.istd1[c];
      k = a.coef1[c], A = -k * mx * is, B = k * (mx * is * m - md);
    }
    s_k[0][c] = k, s_k[1][c] = A, s_k[2][c] = B;
  }
  __syncthreads();
  float* T = s_t[wid];
  const int nt1 = NT1 ? NT1 : a.C1 / 16, G = a.ns / 16;
  st_f4 wx[ST_NT], wy[ST_NT], wz[ST_NT];
#pragma unroll
  for (int qs = 0; qs < ST_NT; ++qs) wx[qs] = wy[qs] = wz[qs] = (st_f4){0.f, 0.f, 0.f, 0.f};
  for (int64_t q = (int64_t)blockIdx.x * (SG_THREADS / 64) + wid; q < a.M; q += (int64_t)gridDim.x * (SG_THREADS / 64)) {
    const bool empty = a.idx[q * a.ns] < 0;
    const int64_t base = a.row_start[q];
    const float qx = a.new_xyz[q * 3], qy = a.new_xyz[q * 3 + 1], qz = a.new_xyz[q * 3 + 2];
    const int64_t first = empty ? -1 : base + a.idx[q * a.ns];
    float acc0 = 0.f;                                                // repeats of the first neighbour (slots past the ball's count) summed here
    for (int g = 0; g < G; ++g) {
      const int64_t row = q * a.ns + g * 16 + li;
      const int32_t my = a.idx[row];                                 // lane li (any kk) holds tile row li's neighbour
      float dx = 0.f, dy = 0.f, dz = 0.f;
      if (!empty) {
        const float* p = a.xyz + (base + my) * 3;
        dx = p[0] - qx, dy = p[1] - qy, dz = p[2] - qz;
      }
#pragma unroll
      for (int qs = 0; qs < ST_NT; ++qs)
        if (qs < nt1) {
          const int c = qs * 16 + kk * 4;
          const st_f4 d = *reinterpret_cast<const st_f4*>(a.dy1 + row * a.C1 + c), z = *reinterpret_cast<const st_f4*>(a.z1 + row * a.C1 + c);
          const st_f4 k = *reinterpret_cast<const st_f4*>(s_k[0] + c), A = *reinterpret_cast<const st_f4*>(s_k[1] + c),
                      B = *reinterpret_cast<const st_f4*>(s_k[2] + c);
          st_f4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = fmaf(k[i], d[i], fmaf(A[i], z[i], B[i]));
            wx[qs][i] = fmaf(v[i], dx, wx[qs][i]), wy[qs][i] = fmaf(v[i], dy, wy[qs][i]), wz[qs][i] = fmaf(v[i], dz, wz[qs][i]);
          }
          *reinterpret_cast<st_f4*>(T + li * ST_PITCH + c) = v;
        }
      if (a.S && !empty) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (int r = 0; r < 16; ++r) {
          const int64_t src = base + __shfl(my, r);
          const float v = lane < a.C1 ? T[r * ST_PITCH + lane] : 0.f;
          if (src == first) acc0 += v;
          else if (lane < a.C1) atomicAdd(a.S + src * a.C1 + lane, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      }
    }
    if (a.S && !empty && lane < a.C1) atomicAdd(a.S + first * a.C1 + lane, acc0);
  }
  // xyz columns of dW1: over the 16 rows of a wave by shuffles, over the waves in a fixed order -> partial (grid, C1, 4) [x y z -]
#pragma unroll
  for (int qs = 0; qs < ST_NT; ++qs)
    if (qs < nt1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float u = wx[qs][i], v = wy[qs][i], w = wz[qs][i];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) u += __shfl_xor(u, off), v += __shfl_xor(v, off), w += __shfl_xor(w, off);
        if (li == 0) {
          const int c = qs * 16 + kk * 4 + i;
          s_wxr[wid][0][c] = u, s_wxr[wid][1][c] = v, s_wxr[wid][2][c] = w;
        }
      }
    }
  __syncthreads();
  for (int e = tid; e < a.C1 * 4; e += SG_THREADS) {
    const int c = e >> 2, d = e & 3;
    float v = 0.f;
    if (d < 3)
      for (int w = 0; w < SG_THREADS / 64; ++w) v += s_wxr[w][d][c];
    a.wpart[(size_t)blockIdx.x * a.C1 * 4 + e] = v;
  }
}

// wgp: the feature columns of dW1 = S^T . F, a contraction over the N support points on the matrix core.  blockIdx.y = feature tile (16
// columns of F), a wave takes 16 points at a time: A = S tile read column-wise (point 4 kk + s is the k index of MFMA s), B = F tile likewise.
// Partial (gridDim.x, C1, C).
__global__ __launch_bounds__(SP_THREADS) void k_sa_wgrad1_points(SaT a) {
  __shared__ __attribute__((aligned(16))) float s_t[SP_THREADS / 64][16 * ST_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, kk = lane >> 4;
  const int v = blockIdx.y, nt1 = a.C1 / 16;
  st_f4 acc[ST_NT];
#pragma unroll
  for (int t = 0; t < ST_NT; ++t) acc[t] = (st_f4){0.f, 0.f, 0.f, 0.f};
  const int64_t ntiles = (a.N + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * (SP_THREADS / 64) + wid; tile < ntiles; tile += (int64_t)gridDim.x * (SP_THREADS / 64)) {
    st_f4 X = (st_f4){0.f, 0.f, 0.f, 0.f}, Sv[ST_NT];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int64_t n = tile * 16 + kk * 4 + s4;
      if (n < a.N) X[s4] = a.feat[n * a.C + v * 16 + li];
    }
#pragma unroll
    for (int t = 0; t < ST_NT; ++t) {
      Sv[t] = (st_f4){0.f, 0.f, 0.f, 0.f};
      if (t < nt1) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int64_t n = tile * 16 + kk * 4 + s4;
          if (n < a.N) Sv[t][s4] = a.S[n * a.C1 + t * 16 + li];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < ST_NT; ++t)
      if (t < nt1) acc[t] = mfma4(Sv[t], X, acc[t]);
  }
  // acc[t][r] = dW1[c1 = 16 t + 4 kk + r][feature 16 v + li]: sum over the waves, then the workgroup's partial
  float* wp = a.wpart + (size_t)blockIdx.x * a.C1 * a.C;
  float* T = s_t[wid];
#pragma unroll
  for (int t = 0; t < ST_NT; ++t)
    if (t < nt1) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(kk * 4 + r) * ST_PITCH + li] = acc[t][r];
      __syncthreads();
      for (int e = tid; e < 256; e += SP_THREADS) {
        const int row = e >> 4, col = e & 15;
        float s4 = 0.f;
        for (int w = 0; w < SP_THREADS / 64; ++w) s4 += s_t[w][row * ST_PITCH + col];
        wp[(t * 16 + row) * a.C + v * 16 + col] = s4;
      }
    }
}

// grad_features (N, C) = scatter (N, C1) . W1[:, 3:]: the gathered feature row enters layer 1 linearly, so its gradient is the per-point sum of
// dz1 times the feature columns of the layer-1 weight
__global__ __launch_bounds__(256) void k_sa_feat_grad(const float* __restrict__ S, const float* __restrict__ w1, int64_t N, int C, int C1,
                                                      float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float s_w[ST_MAXC * ST_MAXF];          // (C1, C)
  const int tid = threadIdx.x, c4n = C >> 2, rpb = 256 / c4n;                // C in {16 .. 128}: 4 .. 32 threads per row
  for (int e = tid; e < C1 * C; e += 256) s_w[e] = w1[(e / C) * (C + 3) + 3 + e % C];
  __syncthreads();
  const int c4 = tid % c4n, rr = tid / c4n;
  if (rr >= rpb) return;
  for (int64_t n = (int64_t)blockIdx.x * rpb + rr; n < N; n += (int64_t)gridDim.x * rpb) {
    const float* srow = S + n * C1;
    st_f4 acc = (st_f4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < C1; k += 4) {
      const st_f4 sv = *reinterpret_cast<const st_f4*>(srow + k);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const st_f4 w = *reinterpret_cast<const st_f4*>(s_w + (k + j) * C + c4 * 4);
        acc.x = fmaf(sv[j], w.x, acc.x), acc.y = fmaf(sv[j], w.y, acc.y), acc.z = fmaf(sv[j], w.z, acc.z), acc.w = fmaf(sv[j], w.w, acc.w);
      }
    }
    reinterpret_cast<st_f4*>(out + n * C)[c4] = acc;
  }
}

// sum of the workgroup partials in a fixed order: out[n * out_stride + out_col0 + k] = sum over p of part[(p * rows + n) * cols_p + k], k < ncols.
// 16 elements per block, 16 groups of partials per element, combined in LDS in a fixed order.
__global__ __launch_bounds__(256) void k_sa_wreduce(const float* __restrict__ part, int nparts, int rows, int cols_p, int ncols, float* __restrict__ out,
                                                    int out_stride, int out_col0) {
  __shared__ float s_r[16][16];
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float v = 0.f;
  const int n = e / ncols, k = e - n * ncols;
  if (e < rows * ncols) {
    const int per = (nparts + 15) / 16, p0 = grp * per, p1 = min(nparts, p0 + per);
    for (int p = p0; p < p1; ++p) v += part[((size_t)p * rows + n) * cols_p + k];
  }
  s_r[grp][el] = v;
  __syncthreads();
  if (grp == 0 && e < rows * ncols) {
    float t = 0.f;
    for (int g = 0; g < 16; ++g) t += s_r[g][el];
    out[(size_t)n * out_stride + out_col0 + k] = t;
  }
}

// forward scale / shift of a BatchNorm from its saved batch statistics: o = {gamma * invstd, beta - mean * gamma * invstd}
__global__ void k_sa_coef(const float* gamma, const float* beta, const float* mean, const float* istd, int C, float* o) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  const float sc = istd[c] * g;                 // exactly k_bn_finalize's forward expressions (norm.h): the ReLU-1 mask of the backward is recomputed from them
  o[c] = sc;
  o[C + c] = bn_shift(mean[c], sc, bt);
}

// ------------------------------------------------------------------------------------------------ host side
static int sa_grid(int64_t M, int per_wg) {
  const int64_t need = (M + per_wg - 1) / per_wg;
  return (int)(need < ST_GRID ? (need < 1 ? 1 : need) : ST_GRID);
}

static int sa_train_check(const char* who, int64_t M, int C, int ns, int C1, int C2) {
  SV_CHECK_ARG(M >= 1, "%s: needs at least one query", who);
  SV_CHECK_ARG(C >= 0 && C % 16 == 0 && C <= ST_MAXF, "%s: feature channels must be a multiple of 16 up to %d (got %d)", who, ST_MAXF, C);
  SV_CHECK_ARG(ns == 16 || ns == 32, "%s: nsample must be 16 or 32 (got %d)", who, ns);
  SV_CHECK_ARG(C1 % 16 == 0 && C2 % 16 == 0 && C1 >= 16 && C2 >= 16 && C1 <= ST_MAXC && C2 <= ST_MAXC,
               "%s: MLP channels must be 16, 32, 48 or 64 (got %d, %d)", who, C1, C2);
  SV_CHECK_ARG(M * (int64_t)ns >= 2, "%s: BatchNorm needs more than one value per channel", who);
  return SV_OK;
}

// grids: fwd1 / bwd1 (gather passes, 256 threads, no matrix work) up to SG_GRID workgroups; fwd2 / bwd2 up to ST_GRID; statistics partials sized for
// the larger of the two
constexpr int SG_GRID = 1024;
constexpr int WGP_GRID = 64;
static int sg_grid(int64_t M) {
  const int64_t need = (M + SG_THREADS / 64 - 1) / (SG_THREADS / 64);
  return (int)(need < SG_GRID ? (need < 1 ? 1 : need) : SG_GRID);
}

// scratch layout (floats): coef1 (4 C1) | coef2 (4 C2) | stats partials (SG_GRID * 2 * 64) | weight partials
extern "C" size_t sv_sa_train_scratch_bytes(int C, int C1, int C2) {
  const size_t w2 = (size_t)ST_GRID * C2 * C1, wx = (size_t)SG_GRID * C1 * 4, wf = (size_t)WGP_GRID * C1 * (C > 0 ? C : 1);
  const size_t w = w2 > wx ? (w2 > wf ? w2 : wf) : (wx > wf ? wx : wf);
  return (4 * (size_t)(C1 + C2) + (size_t)SG_GRID * 2 * ST_MAXC + w) * sizeof(float);
}

struct SaScratch {
  float *coef1, *coef2, *part, *wpart;
};
static SaScratch sa_scratch(void* scratch, int C1, int C2) {
  SaScratch s;
  s.coef1 = reinterpret_cast<float*>(scratch);
  s.coef2 = s.coef1 + 4 * C1;
  s.part = s.coef2 + 4 * C2;
  s.wpart = s.part + (size_t)SG_GRID * 2 * ST_MAXC;
  return s;
}

extern "C" int sv_sa_train_forward(const float* xyz, const float* features, const float* new_xyz, const int32_t* idx, const int32_t* row_start,
                                   int64_t M, int64_t N, int C, int nsample, const float* w1, const float* gamma1, const float* beta1, float* running_mean1,
                                   float* running_var1, int64_t* tracked1, int C1, const float* w2, const float* gamma2, const float* beta2,
                                   float* running_mean2, float* running_var2, int64_t* tracked2, int C2, float momentum, float eps, void* scratch,
                                   float* proj, float* z1, float* z2, float* save_mean1, float* save_invstd1, float* save_mean2, float* save_invstd2,
                                   float* sel, float* aux, uint8_t* arg, uint8_t* aux_arg, float* out, void* stream) {
  if (int rc = sa_train_check("sv_sa_train_forward", M, C, nsample, C1, C2)) return rc;
  SV_CHECK_ARG(xyz && new_xyz && idx && row_start && w1 && w2 && scratch && z1 && z2 && save_mean1 && save_invstd1 && save_mean2 && save_invstd2 &&
                   sel && aux && arg && aux_arg && out && ((features && proj) || C == 0),
               "sv_sa_train_forward: null pointer");
  SV_CHECK_ARG(N >= 0, "sv_sa_train_forward: bad point count");
  SV_CHECK_ARG(C == 0 || (uintptr_t)features % 16 == 0, "sv_sa_train_forward: features must be 16-byte aligned");
  hipStream_t st = sv_stream(stream);
  const SaScratch sc = sa_scratch(scratch, C1, C2);
  const int64_t R = M * nsample;
  SaT a{};
  a.xyz = xyz, a.feat = C ? features : nullptr, a.new_xyz = new_xyz, a.idx = idx, a.row_start = row_start, a.w1 = w1, a.w2 = w2, a.z1 = z1, a.z2 = z2;
  a.part = sc.part, a.coef1 = sc.coef1, a.coef2 = sc.coef2, a.zmax = sel, a.zmin = aux, a.amax = arg, a.amin = aux_arg, a.out = out;
  a.M = M, a.N = N, a.C = C, a.Kp = 16 * (C / 16 + 1), a.C1 = C1, a.C2 = C2, a.ns = nsample;
  a.P = C ? proj : nullptr;
  if (C && N > 0) hipLaunchKernelGGL(k_sa_point_proj, dim3(sv_grid_1d((N + 15) / 16, SP_THREADS / 64, 1024)), dim3(SP_THREADS), 0, st, a);
  const int gridg = sg_grid(M);
  if (C1 == 64 && C2 == 64) hipLaunchKernelGGL((k_sa_fwd1<4, 4>), dim3(gridg), dim3(SG_THREADS), 0, st, a);
  else if (C1 == 32 && C2 == 32) hipLaunchKernelGGL((k_sa_fwd1<2, 2>), dim3(gridg), dim3(SG_THREADS), 0, st, a);
  else if (C1 == 16 && C2 == 16) hipLaunchKernelGGL((k_sa_fwd1<1, 1>), dim3(gridg), dim3(SG_THREADS), 0, st, a);
  else hipLaunchKernelGGL((k_sa_fwd1<0, 0>), dim3(gridg), dim3(SG_THREADS), 0, st, a);
  BnArgs b{};
  b.gamma = gamma1, b.beta = beta1, b.running_mean = running_mean1, b.running_var = running_var1, b.save_mean = save_mean1, b.save_invstd = save_invstd1;
  b.partial = sc.part, b.coef = sc.coef1, b.n = R, b.C = C1, b.wgs = gridg, b.momentum = momentum, b.eps = eps, b.num_batches_tracked = tracked1;
  sv_bn_finalize_fwd(b, st);
  const int grid = sa_grid(M, ST_WAVES);
  if (C1 == 64 && C2 == 64) hipLaunchKernelGGL((k_sa_fwd2<4, 4>), dim3(grid), dim3(ST_THREADS), 0, st, a);
  else if (C1 == 32 && C2 == 32) hipLaunchKernelGGL((k_sa_fwd2<2, 2>), dim3(grid), dim3(ST_THREADS), 0, st, a);
  else if (C1 == 16 && C2 == 16) hipLaunchKernelGGL((k_sa_fwd2<1, 1>), dim3(grid), dim3(ST_THREADS), 0, st, a);
  else hipLaunchKernelGGL((k_sa_fwd2<0, 0>), dim3(grid), dim3(ST_THREADS), 0, st, a);
  b.wgs = grid;
  b.gamma = gamma2, b.beta = beta2, b.running_mean = running_mean2, b.running_var = running_var2, b.save_mean = save_mean2, b.save_invstd = save_invstd2;
  b.coef = sc.coef2, b.C = C2, b.num_batches_tracked = tracked2;
  sv_bn_finalize_fwd(b, st);
  hipLaunchKernelGGL(k_sa_out, dim3(sv_grid_1d(M * C2, 256)), dim3(256), 0, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// grad_out (M, C2) -> grad_w1 (C1, 3 + C), grad_w2 (C2, C1), dgamma / dbeta of both norms, scatter (N, C1) = sum over a support point's
// (query, slot) pairs of dz1 (zeroed here), grad_features (N, C) = scatter . w1[:, 3:] (null: not wanted).  z1 / z2 / sel / arg / out /
// save_* as the forward left them; dy1 (R, C1) and aux (M, C2) are work buffers.
extern "C" int sv_sa_train_backward(const float* xyz, const float* features, const float* new_xyz, const int32_t* idx, const int32_t* row_start,
                                    int64_t M, int64_t N, int C, int nsample, const float* w1, const float* gamma1, const float* beta1, int C1,
                                    const float* w2, const float* gamma2, const float* beta2, int C2, const float* z1, const float* z2,
                                    const float* save_mean1, const float* save_invstd1, const float* save_mean2, const float* save_invstd2,
                                    const float* sel, const uint8_t* arg, const float* out, const float* grad_out, void* scratch, float* dy1,
                                    float* aux, float* scatter, float* grad_features, float* grad_w1, float* grad_w2, float* dgamma1, float* dbeta1,
                                    float* dgamma2, float* dbeta2, void* stream) {
  if (int rc = sa_train_check("sv_sa_train_backward", M, C, nsample, C1, C2)) return rc;
  SV_CHECK_ARG(xyz && new_xyz && idx && row_start && w1 && w2 && z1 && z2 && save_mean1 && save_invstd1 && save_mean2 && save_invstd2 && sel && arg &&
                   out && grad_out && scratch && dy1 && aux && grad_w1 && grad_w2 && dgamma1 && dbeta1 && dgamma2 && dbeta2 &&
                   ((features && scatter) || C == 0),
               "sv_sa_train_backward: null pointer");
  hipStream_t st = sv_stream(stream);
  const SaScratch sc = sa_scratch(scratch, C1, C2);
  const int64_t R = M * nsample;
  SaT a{};
  a.xyz = xyz, a.feat = C ? features : nullptr, a.new_xyz = new_xyz, a.idx = idx, a.row_start = row_start, a.w1 = w1, a.w2 = w2;
  a.z1 = const_cast<float*>(z1), a.z2 = const_cast<float*>(z2), a.dy1 = dy1, a.part = sc.part, a.coef1 = sc.coef1, a.coef2 = sc.coef2;
  a.istd1 = save_invstd1, a.mean1 = save_mean1, a.istd2 = save_invstd2, a.mean2 = save_mean2;
  a.zmax = const_cast<float*>(sel), a.zmin = aux, a.amax = const_cast<uint8_t*>(arg), a.out = const_cast<float*>(out), a.dout = grad_out;
  a.wpart = sc.wpart, a.S = C ? scatter : nullptr, a.M = M, a.N = N, a.C = C, a.Kp = 16 * (C / 16 + 1), a.C1 = C1, a.C2 = C2, a.ns = nsample;
  // BatchNorm 2 backward sums
  const int g0 = sa_grid(M, 64);
  hipLaunchKernelGGL(k_sa_bwd0, dim3(g0), dim3(256), 0, st, a);
  BnArgs b{};
  b.gamma = gamma2, b.beta = beta2, b.save_mean = const_cast<float*>(save_mean2), b.save_invstd = const_cast<float*>(save_invstd2);
  b.dgamma = dgamma2, b.dbeta = dbeta2, b.partial = sc.part, b.coef = sc.coef2, b.n = R, b.C = C2, b.wgs = g0;
  sv_bn_finalize_bwd(b, st);
  // forward scale / shift of BatchNorm 1 (the ReLU mask and a1 are recomputed from z1)
  hipLaunchKernelGGL(k_sa_coef, dim3(1), dim3(64), 0, st, gamma1, beta1, save_mean1, save_invstd1, C1, sc.coef1);
  const int grid = sa_grid(M, 4);
  if (C1 == 64 && C2 == 64) hipLaunchKernelGGL((k_sa_bwd2<4, 4>), dim3(grid), dim3(ST_THREADS), 0, st, a);
  else if (C1 == 32 && C2 == 32) hipLaunchKernelGGL((k_sa_bwd2<2, 2>), dim3(grid), dim3(ST_THREADS), 0, st, a);
  else if (C1 == 16 && C2 == 16) hipLaunchKernelGGL((k_sa_bwd2<1, 1>), dim3(grid), dim3(ST_THREADS), 0, st, a);
  else hipLaunchKernelGGL((k_sa_bwd2<0, 0>), dim3(grid), dim3(ST_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_sa_wreduce, dim3(sv_div_up((int64_t)C2 * C1, 16)), dim3(256), 0, st, sc.wpart, grid, C2, C1, C1, grad_w2, C1, 0);
  // BatchNorm 1 backward
  b.gamma = gamma1, b.beta = beta1, b.save_mean = const_cast<float*>(save_mean1), b.save_invstd = const_cast<float*>(save_invstd1);
  b.dgamma = dgamma1, b.dbeta = dbeta1, b.coef = sc.coef1, b.C = C1, b.wgs = grid;
  sv_bn_finalize_bwd(b, st);
  if (C) SV_HIP(hipMemsetAsync(scatter, 0, (size_t)N * C1 * sizeof(float), st));
  const int grid1 = sg_grid(M);
  if (C1 == 64 && C2 == 64) hipLaunchKernelGGL((k_sa_bwd1<4, 4>), dim3(grid1), dim3(SG_THREADS), 0, st, a);
  else if (C1 == 32 && C2 == 32) hipLaunchKernelGGL((k_sa_bwd1<2, 2>), dim3(grid1), dim3(SG_THREADS), 0, st, a);
  else if (C1 == 16 && C2 == 16) hipLaunchKernelGGL((k_sa_bwd1<1, 1>), dim3(grid1), dim3(SG_THREADS), 0, st, a);
  else hipLaunchKernelGGL((k_sa_bwd1<0, 0>), dim3(grid1), dim3(SG_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_sa_wreduce, dim3(sv_div_up((int64_t)C1 * 3, 16)), dim3(256), 0, st, sc.wpart, grid1, C1, 4, 3, grad_w1, C + 3, 0);      // xyz columns
  if (C) {                                                                                                                        // feature columns
    const int gp = sv_grid_1d((N + 15) / 16, SP_THREADS / 64, WGP_GRID);
    hipLaunchKernelGGL(k_sa_wgrad1_points, dim3(gp, C / 16), dim3(SP_THREADS), 0, st, a);
    hipLaunchKernelGGL(k_sa_wreduce, dim3(sv_div_up((int64_t)C1 * C, 16)), dim3(256), 0, st, sc.wpart, gp, C1, C, C, grad_w1, C + 3, 3);
  }
  if (C && grad_features) hipLaunchKernelGGL(k_sa_feat_grad, dim3(sv_grid_1d(N, 256 / (C / 4), 1024)), dim3(256), 0, st, scatter, w1, N, C, C1, grad_features);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
