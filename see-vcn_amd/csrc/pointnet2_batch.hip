// PointNet++ batch-layout primitives and the 3-NN interpolation of both layouts — the remaining wrappers of the reference's
// pointnet2_batch_cuda / pointnet2_stack_cuda extensions (detector3d/pcdet/ops/pointnet2/pointnet2_batch/src/*.cu,
// pointnet2_stack/src/interpolate_gpu.cu; pybind names in pointnet2_api.cpp).  Same results as the reference kernels
// (scan order, strict comparisons, fp32 arithmetic without contraction); gradient kernels use float atomics like theirs and
// zero-fill their output themselves.
#include "common.h"

#define PB_TILE 512

// ball_query_kernel_fast (ball_query_gpu.cu:13-48): first `nsample` points with d2 < r2 in index order; the first hit fills every slot
__global__ __launch_bounds__(256) void k_ball_query_batch(int n, int m, float radius2, int nsample, const float* __restrict__ new_xyz,
                                                        const float* __restrict__ xyz, int32_t* __restrict__ idx) {
  __shared__ float buf[PB_TILE * 3];
  const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  const bool live = q < m;
  const float* p = new_xyz + ((size_t)b * m + (live ? q : 0)) * 3;
  const float nx = p[0], ny = p[1], nz = p[2];
  int32_t* out = idx + ((size_t)b * m + (live ? q : 0)) * nsample;
  int cnt = live ? 0 : nsample;
  for (int k0 = 0; k0 < n; k0 += PB_TILE) {
    const int c = min(n - k0, PB_TILE);
    __syncthreads();
    for (int t = threadIdx.x; t < c * 3; t += 256) buf[t] = xyz[((size_t)b * n + k0) * 3 + t];
    __syncthreads();
    if (cnt >= nsample) continue;
    for (int k = 0; k < c; ++k) {
      const float dx = nx - buf[k * 3], dy = ny - buf[k * 3 + 1], dz = nz - buf[k * 3 + 2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < radius2) {
        if (cnt == 0)
          for (int l = 0; l < nsample; ++l) out[l] = k0 + k;
        out[cnt] = k0 + k;
        if (++cnt >= nsample) break;
      }
    }
  }
}

// group_points_kernel_fast / gather_points_kernel_fast: out[b][c][j] = points[b][c][idx[b][j]]   (j over npoint*nsample)
__global__ __launch_bounds__(256) void k_gather_cn(int c, int n, int64_t per_batch, const float* __restrict__ points,
                                                 const int32_t* __restrict__ idx, float* __restrict__ out) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= per_batch) return;
  out[((size_t)b * c + ch) * per_batch + j] = points[((size_t)b * c + ch) * n + idx[(size_t)b * per_batch + j]];
}

__global__ __launch_bounds__(256) void k_gather_cn_grad(int c, int n, int64_t per_batch, const float* __restrict__ grad_out,
                                                      const int32_t* __restrict__ idx, float* __restrict__ grad_points) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= per_batch) return;
  atomicAdd(&grad_points[((size_t)b * c + ch) * n + idx[(size_t)b * per_batch + j]], grad_out[((size_t)b * c + ch) * per_batch + j]);
}

// three_nn_kernel_fast / three_nn_kernel_stack: squared distances in float, running bests in double (1e40), strict '<' cascade.
// `known` points of the query's batch element are [k_start, k_start + k_cnt); returned indices are k_start-relative + idx_base.
__device__ __forceinline__ void three_nn_scan(float ux, float uy, float uz, const float* __restrict__ known, int k_cnt, int idx_base,
                                              float* dist2, int32_t* idx) {
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int b1 = 0, b2 = 0, b3 = 0;
  for (int k = 0; k < k_cnt; ++k) {
    const float x = known[k * 3], y = known[k * 3 + 1], z = known[k * 3 + 2];
    const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
    if (d < best1) {
      best3 = best2, b3 = b2;
      best2 = best1, b2 = b1;
      best1 = d, b1 = k;
    } else if (d < best2) {
      best3 = best2, b3 = b2;
      best2 = d, b2 = k;
    } else if (d < best3) {
      best3 = d, b3 = k;
    }
  }
  dist2[0] = (float)best1, dist2[1] = (float)best2, dist2[2] = (float)best3;
  idx[0] = b1 + idx_base, idx[1] = b2 + idx_base, idx[2] = b3 + idx_base;
}

__global__ __launch_bounds__(256) void k_three_nn_batch(int n, int m, const float* __restrict__ unknown, const float* __restrict__ known,
                                                      float* __restrict__ dist2, int32_t* __restrict__ idx) {
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const float* u = unknown + ((size_t)b * n + p) * 3;
  three_nn_scan(u[0], u[1], u[2], known + (size_t)b * m * 3, m, 0, dist2 + ((size_t)b * n + p) * 3, idx + ((size_t)b * n + p) * 3);
}

// three_interpolate: batch layout points (B,C,M) -> out (B,C,N)
__global__ __launch_bounds__(256) void k_three_interp_batch(int c, int m, int n, const float* __restrict__ points, const int32_t* __restrict__ idx,
                                                          const float* __restrict__ w, float* __restrict__ out) {
  const int b = blockIdx.z, ch = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int32_t* i3 = idx + ((size_t)b * n + p) * 3;
  const float* w3 = w + ((size_t)b * n + p) * 3;
  const float* src = points + ((size_t)b * c + ch) * m;
  out[((size_t)b * c + ch) * n + p] = w3[0] * src[i3[0]] + w3[1] * src[i3[1]] + w3[2] * src[i3[2]];
}

__global__ __launch_bounds__(256) void k_three_interp_grad_batch(int c, int n, int m, const float* __restrict__ grad_out,
                                                               const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                               float* __restrict__ grad_points) {
  const int b = blockIdx.z, ch = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int32_t* i3 = idx + ((size_t)b * n + p) * 3;
  const float* w3 = w + ((size_t)b * n + p) * 3;
  const float g = grad_out[((size_t)b * c + ch) * n + p];
  float* dst = grad_points + ((size_t)b * c + ch) * m;
  atomicAdd(&dst[i3[0]], g * w3[0]), atomicAdd(&dst[i3[1]], g * w3[1]), atomicAdd(&dst[i3[2]], g * w3[2]);
}

extern "C" int sv_ball_query_batch(int batch, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int32_t* idx,
                                   void* stream) {
  SV_CHECK_ARG(batch >= 0 && n >= 0 && m >= 0 && nsample >= 1, "sv_ball_query_batch: bad sizes");
  if (batch == 0 || m == 0) return SV_OK;
  SV_CHECK_ARG(new_xyz && idx && (xyz || n == 0), "sv_ball_query_batch: null pointer");
  hipLaunchKernelGGL(k_ball_query_batch, dim3(sv_div_up(m, 256), batch), dim3(256), 0, sv_stream(stream), n, m, radius * radius, nsample,
                     new_xyz, xyz, idx);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_group_points_batch(int batch, int c, int n, int npoints, int nsample, const float* points, const int32_t* idx, float* out,
                                     void* stream) {
  const int64_t per = (int64_t)npoints * nsample;
  if (batch <= 0 || c <= 0 || per <= 0) return SV_OK;
  SV_CHECK_ARG(points && idx && out, "sv_group_points_batch: null pointer");
  hipLaunchKernelGGL(k_gather_cn, dim3(sv_div_up(per, 256), c, batch), dim3(256), 0, sv_stream(stream), c, n, per, points, idx, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_group_points_grad_batch(int batch, int c, int n, int npoints, int nsample, const float* grad_out, const int32_t* idx,
                                          float* grad_points, void* stream) {
  const int64_t per = (int64_t)npoints * nsample;
  if (batch <= 0 || c <= 0 || n <= 0) return SV_OK;
  SV_CHECK_ARG(grad_points && (per == 0 || (grad_out && idx)), "sv_group_points_grad_batch: null pointer");
  SV_HIP(hipMemsetAsync(grad_points, 0, (size_t)batch * c * n * sizeof(float), sv_stream(stream)));
  if (per == 0) return SV_OK;
  hipLaunchKernelGGL(k_gather_cn_grad, dim3(sv_div_up(per, 256), c, batch), dim3(256), 0, sv_stream(stream), c, n, per, grad_out, idx,
                     grad_points);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_gather_points_batch(int batch, int c, int n, int npoints, const float* points, const int32_t* idx, float* out,
                                      void* stream) {
  return sv_group_points_batch(batch, c, n, npoints, 1, points, idx, out, stream);
}

extern "C" int sv_gather_points_grad_batch(int batch, int c, int n, int npoints, const float* grad_out, const int32_t* idx,
                                           float* grad_points, void* stream) {
  return sv_group_points_grad_batch(batch, c, n, npoints, 1, grad_out, idx, grad_points, stream);
}

extern "C" int sv_three_nn_batch(int batch, int n, int m, const float* unknown, const float* known, float* dist2, int32_t* idx,
                                 void* stream) {
  if (batch <= 0 || n <= 0) return SV_OK;
  SV_CHECK_ARG(unknown && dist2 && idx && (known || m == 0), "sv_three_nn_batch: null pointer");
  hipLaunchKernelGGL(k_three_nn_batch, dim3(sv_div_up(n, 256), batch), dim3(256), 0, sv_stream(stream), n, m, unknown, known, dist2, idx);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_three_interpolate_batch(int batch, int c, int m, int n, const float* points, const int32_t* idx, const float* weight,
                                          float* out, void* stream) {
  if (batch <= 0 || c <= 0 || n <= 0) return SV_OK;
  SV_CHECK_ARG(points && idx && weight && out, "sv_three_interpolate_batch: null pointer");
  hipLaunchKernelGGL(k_three_interp_batch, dim3(sv_div_up(n, 256), c, batch), dim3(256), 0, sv_stream(stream), c, m, n, points, idx, weight,
                     out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_three_interpolate_grad_batch(int batch, int c, int n, int m, const float* grad_out, const int32_t* idx,
                                               const float* weight, float* grad_points, void* stream) {
  if (batch <= 0 || c <= 0 || m <= 0) return SV_OK;
  SV_CHECK_ARG(grad_points && (n == 0 || (grad_out && idx && weight)), "sv_three_interpolate_grad_batch: null pointer");
  SV_HIP(hipMemsetAsync(grad_points, 0, (size_t)batch * c * m * sizeof(float), sv_stream(stream)));
  if (n <= 0) return SV_OK;
  hipLaunchKernelGGL(k_three_interp_grad_batch, dim3(sv_div_up(n, 256), c, batch), dim3(256), 0, sv_stream(stream), c, n, m, grad_out, idx,
                     weight, grad_points);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
