// Calibration: sustained rate of the fp32-input MFMAs on gfx950 (independent accumulators, operands in registers).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the same loop on RANDOM operands (four values per lane, rotated): the clock a chip holds under MFMA load depends on the data toggling
// (MI355X_MICROARCH.md "DVFS give-back"); kernels on real data are priced against THIS rate, not the one on constants
template <int NACC>
__global__ __launch_bounds__(256) void k16r(float* out, const float* rnd, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) a[i] = rnd[(t * 8 + i) & 0xfffff], b[i] = rnd[(t * 8 + 4 + i) & 0xfffff];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[t] = s;
}
template <typename F>
void run(const char* name, F launch, double flop_per_mfma, int nacc, int blocks, int iters) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  launch(); hipDeviceSynchronize();
  hipEventRecord(s); launch(); hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  double mf = (double)blocks * 4 * iters * nacc;
  printf("%-28s blocks=%4d  %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz, %d waves/SIMD)\n", name, blocks, ms, mf * flop_per_mfma / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (mf / 1024.0 / 1.0), blocks / 256);
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 20000;
  for (int blocks : {256, 512, 1024}) {
    run("16x16x4 f32, 1 acc (dep chain)", [&] { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 2048, 1, blocks, iters);
    run("16x16x4 f32, 4 acc", [&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 2048, 4, blocks, iters);
    run("16x16x4 f32, 16 acc", [&] { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 2048, 16, blocks, iters);
    run("32x32x2 f32, 1 acc (dep chain)", [&] { hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4096, 1, blocks, iters);
    run("32x32x2 f32, 4 acc", [&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4096, 4, blocks, iters);
  }
  // random operands, sustained: 2 s of back-to-back launches first, then the timed one
  float* rnd; hipMalloc(&rnd, (1 << 20) * 4);
  {
    float* h = (float*)malloc((1 << 20) * 4);
    unsigned x = 12345u;
    for (int i = 0; i < (1 << 20); ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xffff) / 65536.0f - 0.5f; }
    hipMemcpy(rnd, h, (1 << 20) * 4, hipMemcpyHostToDevice);
    free(h);
  }
  for (int rep = 0; rep < 60; ++rep) hipLaunchKernelGGL(k16r<16>, dim3(1024), dim3(256), 0, 0, out, rnd, iters);
  hipDeviceSynchronize();
  run("16x16x4 f32, 16 acc, RANDOM", [&] { hipLaunchKernelGGL(k16r<16>, dim3(1024), dim3(256), 0, 0, out, rnd, iters); }, 2048, 16, 1024, iters);
  run("16x16x4 f32, 16 acc, RANDOM", [&] { hipLaunchKernelGGL(k16r<16>, dim3(1024), dim3(256), 0, 0, out, rnd, iters); }, 2048, 16, 1024, iters);
  for (int rep = 0; rep < 60; ++rep) hipLaunchKernelGGL(k16<16>, dim3(1024), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  run("16x16x4 f32, 16 acc, const (sustained)", [&] { hipLaunchKernelGGL(k16<16>, dim3(1024), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 2048, 16, 1024, iters);
  return 0;
}
