// Chamfer distance (nearest neighbour both ways + its gradient): the `chamfer` extension of the reference's VCN training loss
// (see/surface_completion/models/vcn/extensions/chamfer_dist/chamfer.cu:15-201, bound at chamfer_cuda.cpp:36-39).
// dist[b][j] = min_k |xyz1[b][j] - xyz2[b][k]|^2 (fp32, (dx*dx + dy*dy) + dz*dz), idx = first k attaining it.
#include "common.h"

#define CH_TILE 512

__global__ __launch_bounds__(256) void k_chamfer_nn(const float* __restrict__ xyz1, int n, const float* __restrict__ xyz2, int m,
                                                  float* __restrict__ dist, int32_t* __restrict__ idx) {
  __shared__ float buf[CH_TILE * 3];
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  const bool live = j < n;
  const float* p = xyz1 + ((size_t)b * n + (live ? j : 0)) * 3;
  const float x1 = p[0], y1 = p[1], z1 = p[2];
  float best = 0.f;
  int best_k = 0;
  for (int k2 = 0; k2 < m; k2 += CH_TILE) {
    const int cnt = min(m - k2, CH_TILE);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * 3; t += 256) buf[t] = xyz2[((size_t)b * m + k2) * 3 + t];
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const float x2 = buf[k * 3] - x1, y2 = buf[k * 3 + 1] - y1, z2 = buf[k * 3 + 2] - z1;
      const float d = x2 * x2 + y2 * y2 + z2 * z2;
      if ((k2 == 0 && k == 0) || d < best) best = d, best_k = k2 + k;      // strict <: the first minimum wins, as in the reference
    }
  }
  if (live) dist[(size_t)b * n + j] = best, idx[(size_t)b * n + j] = best_k;
}

// grad_xyz1[j] += 2 g (p1 - p2[idx]),  grad_xyz2[idx] -= 2 g (p1 - p2[idx])   (chamfer.cu:134-166; float atomics like the reference)
__global__ __launch_bounds__(256) void k_chamfer_grad(const float* __restrict__ xyz1, int n, const float* __restrict__ xyz2, int m,
                                                    const float* __restrict__ grad_dist, const int32_t* __restrict__ idx,
                                                    float* __restrict__ grad1, float* __restrict__ grad2) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const size_t a = ((size_t)b * n + j) * 3;
  const int j2 = idx[(size_t)b * n + j];
  const size_t c = ((size_t)b * m + j2) * 3;
  const float g = grad_dist[(size_t)b * n + j] * 2;
  const float gx = g * (xyz1[a] - xyz2[c]), gy = g * (xyz1[a + 1] - xyz2[c + 1]), gz = g * (xyz1[a + 2] - xyz2[c + 2]);
  atomicAdd(&grad1[a], gx), atomicAdd(&grad1[a + 1], gy), atomicAdd(&grad1[a + 2], gz);
  atomicAdd(&grad2[c], -gx), atomicAdd(&grad2[c + 1], -gy), atomicAdd(&grad2[c + 2], -gz);
}

extern "C" int sv_chamfer_forward(const float* xyz1, const float* xyz2, int batch, int n, int m, float* dist1, float* dist2,
                                  int32_t* idx1, int32_t* idx2, void* stream) {
  SV_CHECK_ARG(batch >= 0 && n >= 1 && m >= 1, "sv_chamfer_forward: bad sizes (batch %d, n %d, m %d)", batch, n, m);
  if (batch == 0) return SV_OK;
  SV_CHECK_ARG(xyz1 && xyz2 && dist1 && dist2 && idx1 && idx2, "sv_chamfer_forward: null pointer");
  hipStream_t st = sv_stream(stream);
  hipLaunchKernelGGL(k_chamfer_nn, dim3(sv_div_up(n, 256), batch), dim3(256), 0, st, xyz1, n, xyz2, m, dist1, idx1);
  hipLaunchKernelGGL(k_chamfer_nn, dim3(sv_div_up(m, 256), batch), dim3(256), 0, st, xyz2, m, xyz1, n, dist2, idx2);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_chamfer_backward(const float* xyz1, const float* xyz2, const int32_t* idx1, const int32_t* idx2,
                                   const float* grad_dist1, const float* grad_dist2, int batch, int n, int m, float* grad_xyz1,
                                   float* grad_xyz2, void* stream) {
  SV_CHECK_ARG(batch >= 0 && n >= 1 && m >= 1, "sv_chamfer_backward: bad sizes");
  if (batch == 0) return SV_OK;
  SV_CHECK_ARG(xyz1 && xyz2 && idx1 && idx2 && grad_dist1 && grad_dist2 && grad_xyz1 && grad_xyz2, "sv_chamfer_backward: null pointer");
  hipStream_t st = sv_stream(stream);
  SV_HIP(hipMemsetAsync(grad_xyz1, 0, (size_t)batch * n * 3 * sizeof(float), st));
  SV_HIP(hipMemsetAsync(grad_xyz2, 0, (size_t)batch * m * 3 * sizeof(float), st));
  hipLaunchKernelGGL(k_chamfer_grad, dim3(sv_div_up(n, 256), batch), dim3(256), 0, st, xyz1, n, xyz2, m, grad_dist1, idx1, grad_xyz1,
                     grad_xyz2);
  hipLaunchKernelGGL(k_chamfer_grad, dim3(sv_div_up(m, 256), batch), dim3(256), 0, st, xyz2, m, xyz1, n, grad_dist2, idx2, grad_xyz2,
                     grad_xyz1);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
