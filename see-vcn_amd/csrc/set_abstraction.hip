// Fused set-abstraction reduction (eval mode): for every query point, gather its ball-query neighbours, run the shared 2-layer MLP
// (1x1 conv + folded BatchNorm + ReLU, twice) on the fp32 matrix core and take the maximum over the neighbours -- one launch per radius
// scale, nothing of size (M, C, nsample) ever touches HBM.
//
// Replaces, for inference, the tail of StackSAModuleMSG.forward (detector3d/pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:78-112):
//   QueryAndGroup (pointnet2_utils.py:112-159: GroupingOperation, xyz made relative to the query, concatenated IN FRONT of the features,
//   an empty ball gives an all-zero group) -> Conv2d 1x1 + BatchNorm2d + ReLU, twice -> max_pool2d over nsample.
// The reference materialises grouped features (M, C+3, ns) with group_points_kernel_stack (group_points_gpu.cu:71-102), then runs cuDNN
// convs over them; PV-RCNN's RoI-grid pool (pvrcnn_head.py:64-109) does this for 27 648 queries per scene with C = 128.
//
// Mapping: a wave owns one query; its nsample neighbours are the 16 rows of one (nsample 16) or two (nsample 32) MFMA tiles
// (v_mfma_f32_16x16x4_f32).  Layer 1 contracts over [features | dx dy dz | zero padding] (the weight columns are re-ordered accordingly by
// sv_sa_prepare_weights, so the gathered feature rows stay 16-byte aligned); its output goes through wave-private LDS to become the A operand
// of layer 2; the maximum over the neighbours is a register max + two cross-lane steps.  Both weight matrices live in LDS for the whole
// launch (workgroups are persistent).  Bound: fp32 MFMA -- 2 * ns * ((C+3) * C1 + C1 * C2) flop per query against 12 + 4 * ns * (C + 1)
// gathered bytes (C = 128, 64/64 channels: 400 kflop for 8.3 KB, 48 flop/B).
#include "common.h"

typedef float sa_f32x4 __attribute__((ext_vector_type(4)));

constexpr int SA_THREADS = 512;            // 8 waves: two per SIMD, one workgroup per CU (LDS)
constexpr int SA_MAX_C = 64;               // widest MLP layer
constexpr int SA_MAX_FEAT = 128;           // most input feature channels (PV-RCNN's RoI-grid pool)
constexpr int SA_MAX_KP = SA_MAX_FEAT + 16;   // padded contraction length of layer 1
constexpr int SA_PITCH_H = SA_MAX_C + 4;   // LDS row pitch of the layer-1 output (bank spread)

struct SaArgs {
  const float* xyz;          // (N, 3) support points
  const float* features;     // (N, C) or null when C == 0
  const float* new_xyz;      // (M, 3) queries
  const int32_t* idx;        // (M, ns) scene-local neighbour indices, raw ball-query output (idx[q][0] = -1: empty ball)
  const int32_t* row_start;  // (M) first support row of the query's scene
  const float* w1;           // (C1, Kp) prepared layer-1 weights: [features (C) | xyz (3) | zeros], BatchNorm scale folded in
  const float* b1;           // (C1) folded BatchNorm shift
  const float* w2;           // (C2, C1)
  const float* b2;           // (C2)
  float* out;                // (M, C2)
  int64_t M;
  int C, Kp, C1, C2, ns;
};

template <int G>   // 16-row tiles per query: nsample = 16 * G
__global__ __launch_bounds__(SA_THREADS) void k_sa_mlp_max(SaArgs a) {
  // static LDS sized for the widest case (a launch may declare up to the CU's 160 KB statically; the dynamic-size attribute is capped at 64 KB):
  // 37.9 + 17.4 + 34.8 * G KB -> one workgroup per CU
  __shared__ __attribute__((aligned(16))) float s_w1[SA_MAX_C * (SA_MAX_KP + 4)];                    // (C1, p1)
  __shared__ __attribute__((aligned(16))) float s_w2[SA_MAX_C * (SA_MAX_C + 4)];                     // (C2, p2)
  __shared__ __attribute__((aligned(16))) float s_h[(SA_THREADS / 64) * G * 16 * SA_PITCH_H];        // (waves, G, 16, SA_PITCH_H)
  const int p1 = a.Kp + 4, p2 = a.C1 + 4;                       // LDS pitches of the weight rows
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, kk = lane >> 4;
  for (int e = tid; e < a.C1 * a.Kp; e += SA_THREADS) s_w1[(e / a.Kp) * p1 + e % a.Kp] = a.w1[e];
  for (int e = tid; e < a.C2 * a.C1; e += SA_THREADS) s_w2[(e / a.C1) * p2 + e % a.C1] = a.w2[e];
  __syncthreads();
  float* h = s_h + (size_t)wid * G * 16 * SA_PITCH_H;
  const int nt1 = a.C1 / 16, nt2 = a.C2 / 16, nq1 = a.Kp / 16, nqf = a.C / 16;
  const int waves = SA_THREADS / 64;

  for (int64_t q = (int64_t)blockIdx.x * waves + wid; q < a.M; q += (int64_t)gridDim.x * waves) {
    // neighbour rows of this lane's tile rows (lane = (row li, channel group kk); tile g holds neighbours 16 g .. 16 g + 15)
    const int32_t first = a.idx[q * a.ns];
    const bool empty = first < 0;                                // wave-uniform
    const int64_t base = a.row_start[q];
    int64_t nrow[G];
#pragma unroll
    for (int g = 0; g < G; ++g) nrow[g] = empty ? -1 : base + a.idx[q * a.ns + g * 16 + li];
    const float qx = a.new_xyz[q * 3], qy = a.new_xyz[q * 3 + 1], qz = a.new_xyz[q * 3 + 2];

    // ---- layer 1: H = relu(X . W1'^T + b1), X = [features | dx dy dz | 0]
    sa_f32x4 acc[G][SA_MAX_C / 16];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int t = 0; t < SA_MAX_C / 16; ++t) acc[g][t] = (sa_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int qs = 0; qs < nq1; ++qs) {
      sa_f32x4 A[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        A[g] = (sa_f32x4){0.f, 0.f, 0.f, 0.f};
        if (nrow[g] >= 0) {
          if (qs < nqf) {
            A[g] = *reinterpret_cast<const sa_f32x4*>(a.features + nrow[g] * a.C + qs * 16 + kk * 4);
          } else if (qs == nqf && kk == 0) {                     // the step that holds the relative coordinates
            const float* p = a.xyz + nrow[g] * 3;
            A[g] = (sa_f32x4){p[0] - qx, p[1] - qy, p[2] - qz, 0.f};
          }
        }
      }
#pragma unroll
      for (int t = 0; t < SA_MAX_C / 16; ++t) {
        if (t < nt1) {
          const sa_f32x4 B = *reinterpret_cast<const sa_f32x4*>(s_w1 + (t * 16 + li) * p1 + qs * 16 + kk * 4);
#pragma unroll
          for (int g = 0; g < G; ++g) {
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].x, B.x, acc[g][t], 0, 0, 0);
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].y, B.y, acc[g][t], 0, 0, 0);
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].z, B.z, acc[g][t], 0, 0, 0);
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].w, B.w, acc[g][t], 0, 0, 0);
          }
        }
      }
    }
    // D layout: col = lane & 15 (+ 16 t), rows 4 kk + r  ->  LDS [row][col], the A layout of layer 2 reads it back row-wise
#pragma unroll
    for (int t = 0; t < SA_MAX_C / 16; ++t) {
      if (t < nt1) {
        const float bias = a.b1[t * 16 + li];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) h[(g * 16 + kk * 4 + r) * SA_PITCH_H + t * 16 + li] = fmaxf(acc[g][t][r] + bias, 0.f);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // wave-private LDS: the wave's own LDS operations complete in order
    // ---- layer 2: Y = relu(H . W2'^T + b2), then the maximum over the neighbours
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int t = 0; t < SA_MAX_C / 16; ++t) acc[g][t] = (sa_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int qs = 0; qs < nt1; ++qs) {
      sa_f32x4 A[G];
#pragma unroll
      for (int g = 0; g < G; ++g) A[g] = *reinterpret_cast<const sa_f32x4*>(h + (g * 16 + li) * SA_PITCH_H + qs * 16 + kk * 4);
#pragma unroll
      for (int t = 0; t < SA_MAX_C / 16; ++t) {
        if (t < nt2) {
          const sa_f32x4 B = *reinterpret_cast<const sa_f32x4*>(s_w2 + (t * 16 + li) * p2 + qs * 16 + kk * 4);
#pragma unroll
          for (int g = 0; g < G; ++g) {
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].x, B.x, acc[g][t], 0, 0, 0);
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].y, B.y, acc[g][t], 0, 0, 0);
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].z, B.z, acc[g][t], 0, 0, 0);
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].w, B.w, acc[g][t], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < SA_MAX_C / 16; ++t) {
      if (t < nt2) {
        float m = -3.4e38f;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[g][t][r]);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        // relu(max(y) + b) == max(relu(y + b)): bias and ReLU after the maximum
        if (kk == 0) a.out[q * a.C2 + t * 16 + li] = fmaxf(m + a.b2[t * 16 + li], 0.f);
      }
    }
  }
}

// Layer-1 weights of the reference layout (C1, 3 + C) [xyz columns first, pointnet2_utils.py:147-152] with the eval-mode BatchNorm folded in
// -> (C1, Kp) [features | xyz | zero padding], Kp = 16 * (C / 16 + 1); b = beta - mean * gamma / sqrt(var + eps).  Layer 2 keeps its layout.
__global__ void k_sa_prepare(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                             const float* __restrict__ mean, const float* __restrict__ var, float eps, int Cout, int Cin, int xyz_first, int Kp,
                             float* __restrict__ w_out, float* __restrict__ b_out) {
  const int n = blockIdx.x;
  if (n >= Cout) return;
  const float sc = gamma[n] / sqrtf(var[n] + eps);
  if (threadIdx.x == 0) b_out[n] = beta[n] - mean[n] * sc;
  for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
    float v = 0.f;
    if (xyz_first) {
      const int C = Cin - 3;
      if (k < C) v = w[(size_t)n * Cin + 3 + k];
      else if (k < C + 3) v = w[(size_t)n * Cin + (k - C)];
    } else if (k < Cin) {
      v = w[(size_t)n * Cin + k];
    }
    w_out[(size_t)n * Kp + k] = v * sc;
  }
}

extern "C" int sv_sa_prepare_weights(const float* weight, const float* bn_weight, const float* bn_bias, const float* running_mean,
                                     const float* running_var, float eps, int c_out, int c_in, int xyz_first, float* w_out, float* b_out,
                                     void* stream) {
  SV_CHECK_ARG(weight && bn_weight && bn_bias && running_mean && running_var && w_out && b_out, "sv_sa_prepare_weights: null pointer");
  SV_CHECK_ARG(c_out > 0 && c_in > 0 && (!xyz_first || c_in >= 3), "sv_sa_prepare_weights: bad sizes");
  const int Kp = xyz_first ? 16 * ((c_in - 3) / 16 + 1) : c_in;
  SV_CHECK_ARG(!xyz_first || (c_in - 3) % 16 == 0, "sv_sa_prepare_weights: feature channels must be a multiple of 16 (got %d)", c_in - 3);
  hipLaunchKernelGGL(k_sa_prepare, dim3(c_out), dim3(64), 0, sv_stream(stream), weight, bn_weight, bn_bias, running_mean, running_var, eps, c_out,
                     c_in, xyz_first, Kp, w_out, b_out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_sa_mlp_max(const float* xyz, const float* features, const float* new_xyz, const int32_t* idx, const int32_t* row_start,
                             int64_t M, int C, int nsample, const float* w1, const float* b1, int C1, const float* w2, const float* b2, int C2,
                             float* out, void* stream) {
  SV_CHECK_ARG(M >= 0 && C >= 0 && C % 16 == 0 && C <= SA_MAX_FEAT, "sv_sa_mlp_max: feature channels must be a multiple of 16 up to %d (got %d)", SA_MAX_FEAT, C);
  SV_CHECK_ARG(nsample == 16 || nsample == 32, "sv_sa_mlp_max: nsample must be 16 or 32 (got %d)", nsample);
  SV_CHECK_ARG(C1 % 16 == 0 && C2 % 16 == 0 && C1 >= 16 && C2 >= 16 && C1 <= SA_MAX_C && C2 <= SA_MAX_C,
               "sv_sa_mlp_max: MLP channels must be 16, 32, 48 or 64 (got %d, %d)", C1, C2);
  if (M == 0) return SV_OK;
  SV_CHECK_ARG(xyz && new_xyz && idx && row_start && w1 && b1 && w2 && b2 && out && (features || C == 0), "sv_sa_mlp_max: null pointer");
  SV_CHECK_ARG(C == 0 || (uintptr_t)features % 16 == 0, "sv_sa_mlp_max: features must be 16-byte aligned");
  SaArgs a{xyz, features, new_xyz, idx, row_start, w1, b1, w2, b2, out, M, C, 16 * (C / 16 + 1), C1, C2, nsample};
  const int G = nsample / 16;
  const int64_t wgs_needed = (M + SA_THREADS / 64 - 1) / (SA_THREADS / 64);
  const int grid = (int)(wgs_needed < 256 ? wgs_needed : 256);
  hipStream_t st = sv_stream(stream);
  if (G == 1) hipLaunchKernelGGL(k_sa_mlp_max<1>, dim3(grid), dim3(SA_THREADS), 0, st, a);
  else hipLaunchKernelGGL(k_sa_mlp_max<2>, dim3(grid), dim3(SA_THREADS), 0, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
