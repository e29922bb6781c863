// How fast does ONE wave issue dependent VALU instructions, and does it depend on how much of the chip is busy?  (the rounds of csrc/fps_bucket.hip are
// a few hundred dependent instructions in a handful of waves: their cost is this number)
// hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_clock valu_clock.hip && /tmp/valu_clock
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_chain(float* out, unsigned long long* ticks, unsigned long long* real, int iters) {
  float v = threadIdx.x * 1e-9f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0, *real = r1 - r0;
  if (v == 12345.f) out[0] = v;
}

int main() {
  float* out;
  unsigned long long *ticks, *real;
  hipMalloc(&out, 4), hipMalloc(&ticks, 8), hipMalloc(&real, 8);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep)
    for (int grid : {1, 4, 32, 256, 2048}) {
      for (int threads : {64, 512}) {
        k_chain<<<grid, threads>>>(out, ticks, real, iters);
        hipEventRecord(a);
        k_chain<<<grid, threads>>>(out, ticks, real, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        unsigned long long t, r;
        hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost), hipMemcpy(&r, real, 8, hipMemcpyDeviceToHost);
        const double n = 64.0 * iters;
        printf("grid %5d x %3d threads: %.2f ns per dependent v_add (wall), %.2f s_memtime ticks per instruction, clock by s_memtime / s_memrealtime = %.0f MHz\n", grid,
               threads, ms * 1e6 / n, t / n, (double)t / r * 100.0);
      }
    }
  return 0;
}
