// PointNet++ stacked-batch primitives used by PV-RCNN's VoxelSetAbstraction / RoI-grid pooling:
// farthest point sampling, ball query, grouping (+ its gradient).
//
// Reference kernels (one thread per query / per element, legacy stream, exit(-1) on error):
//   detector3d/pcdet/ops/pointnet2/pointnet2_stack/src/sampling_gpu.cu:24-140      (FPS)
//   detector3d/pcdet/ops/pointnet2/pointnet2_stack/src/ball_query_gpu.cu:16-66    (ball query)
//   detector3d/pcdet/ops/pointnet2/pointnet2_stack/src/group_points_gpu.cu:15-102 (group / grad)
// Re-designed for 64-wide wavefronts: FPS keeps a scene's points and running distances in registers (no HBM
// traffic inside the M sequential rounds) and needs one barrier per round; ball query is wave-cooperative — a
// wave scans the candidate points 64 at a time with coalesced loads from an LDS tile shared by the workgroup's
// queries and appends hits in index order with ballot + prefix popcount.
#include <stdlib.h>

#include "common.h"

// ------------------------------------------------------------------------------------------------
// Farthest point sampling.  Selection rule reproduced exactly (index-exact, including ties):
//   round j picks argmax_k temp[k] where temp[k] = min over picked p of |x_k - x_p|^2 (temp starts at 1e10),
//   first index is 0.  Ties: the reference reduces per-thread strided maxima (strict '>' keeps the smallest k of a
//   stride class) through a power-of-two tree that keeps the LOWER slot on ties (sampling_gpu.cu:16-21,55-134),
//   i.e. among equal maxima the winner minimises (bitreverse(k mod T), k div T), T = 2^floor(log2 n) <= 1024.
// ------------------------------------------------------------------------------------------------
constexpr int FPS_THREADS = 512;   // 8 waves = 2 per SIMD: 256 VGPRs per lane, so 48 points per thread stay in registers (1024 threads
                                   // capped the kernel at 128 VGPRs: the 16- and 24-point variants spilled ~700 / ~2500 registers to scratch
                                   // and a round took 39 us instead of ~2)
constexpr int FPS_MAXP = 48;       // points per thread held in registers -> n <= 24576 on the register path

__device__ __forceinline__ unsigned long long fps_key(float d, int k, int log2t) {
  // larger key wins: distance first (non-negative floats order as their bit patterns), then the tie rule
  const unsigned int lo = (unsigned int)k & ((1u << log2t) - 1u);
  const unsigned int rev = log2t ? (__brev(lo) >> (32 - log2t)) : 0u;
  const unsigned int tie = (rev << (31 - log2t)) | ((unsigned int)k >> log2t);   // smaller tie = preferred
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0x7FFFFFFFu - tie);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(v, d, 64);
    v = o > v ? o : v;
  }
  return v;
}

// one workgroup per scene; xyz (n,3) of the scene, idx (m) output (scene-local indices, + start when add_offset)
template <int P>
__global__ __launch_bounds__(FPS_THREADS) void k_fps_reg(const float* __restrict__ xyz_all, const int32_t* __restrict__ starts,
                                                         const int32_t* __restrict__ counts, int fixed_n, int m,
                                                         int32_t* __restrict__ idx_all, int add_offset) {
  constexpr int NW = FPS_THREADS / 64;
  __shared__ unsigned long long skey[2][NW];
  __shared__ float sxyz[2][NW][3];
  const int b = blockIdx.x;
  const int start = starts ? starts[b] : b * fixed_n;
  const int n = counts ? counts[b] : fixed_n;
  const float* xyz = xyz_all + (int64_t)start * 3;
  int32_t* idx = idx_all + (int64_t)b * m;
  if (n <= 0 || m <= 0) return;
  int log2t = 0;
  while ((2 << log2t) <= n && log2t < 10) ++log2t;             // T = 2^floor(log2 n), capped at 1024
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float px[P], py[P], pz[P], pt[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int k = tid + i * FPS_THREADS;
    const bool ok = k < n;
    px[i] = ok ? xyz[k * 3] : 0.f;
    py[i] = ok ? xyz[k * 3 + 1] : 0.f;
    pz[i] = ok ? xyz[k * 3 + 2] : 0.f;
    pt[i] = 1e10f;
  }
  if (tid == 0) idx[0] = add_offset ? start : 0;
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];                  // first pick is index 0 (sampling_gpu.cu:44-46)
  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
    // per-thread winner by distance; the tie rule is only evaluated on an exact distance tie, so nothing per point is kept in
    // registers beyond x, y, z and the running minimum distance (hoisted tie keys cost 5 more registers per point and spilled)
    float bd = -1.f, bx = 0.f, by = 0.f, bz = 0.f;
    int bk = -1;
    int tid_r = tid, l2 = log2t;
    asm volatile("" : "+v"(tid_r), "+s"(l2));      // opaque per round: keeps hipcc from hoisting P point indices / tie keys into registers
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int k = tid_r + i * FPS_THREADS;
      // same expression order as the reference (sampling_gpu.cu:62); fp contraction is off for this library
      const float d = (px[i] - x1) * (px[i] - x1) + (py[i] - y1) * (py[i] - y1) + (pz[i] - z1) * (pz[i] - z1);
      const float d2 = fminf(d, pt[i]);
      pt[i] = d2;
      bool up = k < n && d2 > bd;
      if (k < n && d2 == bd) up = fps_key(d2, k, l2) > fps_key(bd, bk, l2);
      bd = up ? d2 : bd;
      bk = up ? k : bk;
      bx = up ? px[i] : bx; by = up ? py[i] : by; bz = up ? pz[i] : bz;
    }
    const unsigned long long best = bk >= 0 ? fps_key(bd, bk, l2) : 0ull;
    const unsigned long long wbest = wave_max_u64(best);
    // the lane that holds the wave's winner publishes key + coordinates (keys are unique per point)
    if (best == wbest && best != 0ull) {
      skey[par][wid] = wbest;
      sxyz[par][wid][0] = bx; sxyz[par][wid][1] = by; sxyz[par][wid][2] = bz;
    } else if (wbest == 0ull && lane == 0) {
      skey[par][wid] = 0ull;                                    // a wave without points must still clear its slot
    }
    __syncthreads();                                            // the only barrier of the round (slots alternate by parity)
    const int sl = lane & (NW - 1);
    const unsigned long long mine = skey[par][sl];
    const unsigned long long w = wave_max_u64(mine);            // every wave reduces the 16 partials redundantly
    const int src = __ffsll((long long)__ballot(mine == w)) - 1;
    x1 = __shfl(sxyz[par][sl][0], src, 64);
    y1 = __shfl(sxyz[par][sl][1], src, 64);
    z1 = __shfl(sxyz[par][sl][2], src, 64);
    if (tid == 0) {
      const unsigned int tie = 0x7FFFFFFFu - (unsigned int)(w & 0xFFFFFFFFull);
      const unsigned int rev = log2t ? (tie >> (31 - log2t)) : 0u;
      const unsigned int lo = log2t ? (__brev(rev) >> (32 - log2t)) : 0u;
      const int old = (int)(((tie & ((1u << (31 - log2t)) - 1u)) << log2t) | lo);
      idx[j] = add_offset ? start + old : old;
    }
  }
}

// streaming variant for scenes too large for the register file: temp lives in HBM (n floats, caller-provided)
__global__ __launch_bounds__(FPS_THREADS) void k_fps_stream(const float* __restrict__ xyz_all, const int32_t* __restrict__ starts,
                                                            const int32_t* __restrict__ counts, int fixed_n, int m,
                                                            float* __restrict__ temp_all, int32_t* __restrict__ idx_all, int add_offset) {
  __shared__ unsigned long long slot[2][FPS_THREADS / 64];
  const int b = blockIdx.x;
  const int start = starts ? starts[b] : b * fixed_n;
  const int n = counts ? counts[b] : fixed_n;
  const float* xyz = xyz_all + (int64_t)start * 3;
  float* temp = temp_all + start;
  int32_t* idx = idx_all + (int64_t)b * m;
  if (n <= 0 || m <= 0) return;
  int log2t = 0;
  while ((2 << log2t) <= n && log2t < 10) ++log2t;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int k = tid; k < n; k += FPS_THREADS) temp[k] = 1e10f;
  if (tid == 0) idx[0] = add_offset ? start : 0;
  int old = 0;
  __syncthreads();
  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
    const float x1 = xyz[old * 3], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
    unsigned long long best = 0ull;
    for (int k = tid; k < n; k += FPS_THREADS) {
      const float x2 = xyz[k * 3], y2 = xyz[k * 3 + 1], z2 = xyz[k * 3 + 2];
      const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
      const float d2 = fminf(d, temp[k]);
      temp[k] = d2;
      const unsigned long long key = fps_key(d2, k, log2t);
      best = key > best ? key : best;
    }
    best = wave_max_u64(best);
    if (lane == 0) slot[par][wid] = best;
    __syncthreads();
    unsigned long long w = slot[par][lane & (FPS_THREADS / 64 - 1)];
    w = wave_max_u64(w);
    const unsigned int tie = 0x7FFFFFFFu - (unsigned int)(w & 0xFFFFFFFFull);
    const unsigned int rev = log2t ? (tie >> (31 - log2t)) : 0u;
    const unsigned int lo = log2t ? (__brev(rev) >> (32 - log2t)) : 0u;
    old = (int)(((tie & ((1u << (31 - log2t)) - 1u)) << log2t) | lo);
    if (tid == 0) idx[j] = add_offset ? start + old : old;
  }
}

// ---- several workgroups per scene.  One workgroup per scene leaves 252 of 256 CUs idle at batch 4 and pays ~4 us per round for its
// 34-40 points per thread (16 ms for the 4096 keypoints of the SEE-VCN PV-RCNN, source-nuscenes/pvrcnn.yaml:111).  Here FPS_W workgroups
// share a scene (2-3 points per thread); every round each publishes its best candidate as two self-tagged 16-byte granules
// {key, x, tag} {y, z, tag} -- one `sc1` store each, no drain, no flag, no fence (MI355X_MICROARCH.md, "data-tagged granules") -- and every wave
// polls the 2 x FPS_W granules of its scene (one per lane, `sc1` loads) until all carry the round's tag.  The records alternate between two
// banks by round parity so that a fast workgroup cannot overwrite a granule a slow one still has to read; the per-wave candidates in LDS
// alternate the same way, which leaves ONE workgroup barrier per round.  Same keys, same tie rule, same winners as the single-workgroup
// kernel.  The FPS_W workgroups of a scene must be resident together (batch * FPS_W <= 256 workgroups, checked by the host); the poll is
// bounded and raises *err instead of hanging.  one_xcd: the scene's workgroups are dealt to ONE XCD (workgroup i runs on XCD i % 8) and
// the granules are plain stores that stay in that XCD's L2, where the `sc1` polls of the same XCD find them: 1.9 us per round instead of
// 2.8 with write-through (`sc1`) granules that any XCD may read (measured, 4 x 17k points -> 4096: 7.9 / 11.3 ms; one workgroup per
// scene 16.4 ms).
constexpr int FPS_W = 16;
struct FpsRecord {
  unsigned int h0[4];          // key lo, key hi, x, tag
  unsigned int h1[4];          // y, z, tag, 0
};
typedef unsigned int fps_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void fps_store16(void* p, fps_u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void fps_store16_l2(void* p, fps_u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ fps_u32x4 fps_load16(const void* p) {
  fps_u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

template <int P>
__global__ __launch_bounds__(FPS_THREADS) void k_fps_multi(const float* __restrict__ xyz_all, const int32_t* __restrict__ starts,
                                                           const int32_t* __restrict__ counts, int fixed_n, int m, int32_t* __restrict__ idx_all,
                                                           int add_offset, FpsRecord* __restrict__ records, unsigned nonce, int32_t* __restrict__ err,
                                                           int batch, int one_xcd, int spin_limit) {
  constexpr int NW = FPS_THREADS / 64;
  static_assert(2 * FPS_W <= 64 && (NW & (NW - 1)) == 0, "one granule per lane");
  __shared__ unsigned long long skey[2][NW];
  __shared__ float sxyz[2][NW][3];
  int b = blockIdx.x / FPS_W, w = blockIdx.x % FPS_W;
  if (one_xcd) {                                                // workgroup i runs on XCD i % 8: a scene's workgroups all on one
    const int slot = blockIdx.x >> 3;
    b = (slot / FPS_W) * 8 + (blockIdx.x & 7);
    w = slot % FPS_W;
    if (b >= batch) return;
  }
  const int start = starts ? starts[b] : b * fixed_n;
  const int n = counts ? counts[b] : fixed_n;
  const float* xyz = xyz_all + (int64_t)start * 3;
  int32_t* idx = idx_all + (int64_t)b * m;
  if (n <= 0 || m <= 0) return;
  int log2t = 0;
  while ((2 << log2t) <= n && log2t < 10) ++log2t;             // T = 2^floor(log2 n), capped at 1024
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  FpsRecord* rec = records + (size_t)b * 2 * FPS_W;            // [parity][workgroup]
  float px[P], py[P], pz[P], pt[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int k = (i * FPS_W + w) * FPS_THREADS + tid;
    const bool ok = k < n;
    px[i] = ok ? xyz[k * 3] : 0.f;
    py[i] = ok ? xyz[k * 3 + 1] : 0.f;
    pz[i] = ok ? xyz[k * 3 + 2] : 0.f;
    pt[i] = 1e10f;
  }
  if (w == 0 && tid == 0) idx[0] = add_offset ? start : 0;
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];                  // first pick is index 0 (sampling_gpu.cu:44-46)
  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
    float bd = -1.f, bx = 0.f, by = 0.f, bz = 0.f;
    int bk = -1;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int k = (i * FPS_W + w) * FPS_THREADS + tid;
      const float d = (px[i] - x1) * (px[i] - x1) + (py[i] - y1) * (py[i] - y1) + (pz[i] - z1) * (pz[i] - z1);
      const float d2 = fminf(d, pt[i]);
      pt[i] = d2;
      bool up = k < n && d2 > bd;
      if (k < n && d2 == bd) up = fps_key(d2, k, log2t) > fps_key(bd, bk, log2t);
      bd = up ? d2 : bd;
      bk = up ? k : bk;
      bx = up ? px[i] : bx; by = up ? py[i] : by; bz = up ? pz[i] : bz;
    }
    const unsigned long long best = bk >= 0 ? fps_key(bd, bk, log2t) : 0ull;
    const unsigned long long wbest = wave_max_u64(best);
    if (best == wbest && best != 0ull) {
      skey[par][wid] = wbest;
      sxyz[par][wid][0] = bx; sxyz[par][wid][1] = by; sxyz[par][wid][2] = bz;
    } else if (wbest == 0ull && lane == 0) {
      skey[par][wid] = 0ull;
    }
    __syncthreads();
    const unsigned tag = (nonce << 16) | (unsigned)j;
    FpsRecord* bank = rec + (size_t)par * FPS_W;
    if (wid == 0) {                                             // this workgroup's candidate -> its record
      const int sl = lane & (NW - 1);
      const unsigned long long mine = skey[par][sl];
      const unsigned long long wg_best = wave_max_u64(mine);
      const int src = __ffsll((long long)__ballot(mine == wg_best)) - 1;
      if (lane == src || lane == src + NW) {                    // two lanes hold the winner (NW < 64): one stores each granule
        fps_u32x4 v;
        if (lane == src) v = fps_u32x4{(unsigned)(wg_best & 0xffffffffull), (unsigned)(wg_best >> 32), __float_as_uint(sxyz[par][sl][0]), tag};
        else v = fps_u32x4{__float_as_uint(sxyz[par][sl][1]), __float_as_uint(sxyz[par][sl][2]), tag, 0u};
        if (one_xcd) fps_store16_l2(lane == src ? (void*)bank[w].h0 : (void*)bank[w].h1, v);
        else fps_store16(lane == src ? (void*)bank[w].h0 : (void*)bank[w].h1, v);
      }
    }
    // every wave: poll the 2 x FPS_W granules of the scene, one per lane
    const bool poller = lane < 2 * FPS_W;
    const int half = lane / FPS_W;
    const void* gp = half == 0 ? (const void*)bank[lane & (FPS_W - 1)].h0 : (const void*)bank[lane & (FPS_W - 1)].h1;
    fps_u32x4 g = fps_u32x4{0u, 0u, 0u, 0u};
    bool ready = !poller;
    int spins = 0;
    while (true) {
      if (!ready) {
        g = fps_load16(gp);
        ready = (half == 0 ? g.w : g.z) == tag;
      }
      if (__ballot(!ready) == 0ull) break;
      if (++spins > spin_limit) {                                // ~ seconds: a missing partner (not co-resident) must not hang the GPU
        if (lane == 0) *err = 1;
        return;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    const unsigned long long key = lane < FPS_W ? ((unsigned long long)g.y << 32) | g.x : 0ull;
    const unsigned long long win = wave_max_u64(key);
    const int wl = __ffsll((long long)__ballot(lane < FPS_W && key == win)) - 1;
    x1 = __uint_as_float(__shfl(g.z, wl, 64));
    y1 = __uint_as_float(__shfl(g.x, wl + FPS_W, 64));
    z1 = __uint_as_float(__shfl(g.y, wl + FPS_W, 64));
    if (w == 0 && tid == 0) {
      const unsigned int tie = 0x7FFFFFFFu - (unsigned int)(win & 0xFFFFFFFFull);
      const unsigned int rev = log2t ? (tie >> (31 - log2t)) : 0u;
      const unsigned int lo = log2t ? (__brev(rev) >> (32 - log2t)) : 0u;
      const int old = (int)(((tie & ((1u << (31 - log2t)) - 1u)) << log2t) | lo);
      idx[j] = add_offset ? start + old : old;
    }
  }
}

// async_mode < 0: the multi-workgroup path ends synchronised (error word read back, retry with write-through records).  async_mode 0 / 1: ONE
// attempt (0: one-XCD records, 1: write-through records), nothing is read back -- the caller reads the error word at the end of multi_scratch
// (sv_fps_multi_error_offset) whenever it next synchronises and re-runs with mode 1 if it is non-zero.
static int fps_launch(const float* xyz, const int32_t* starts, const int32_t* counts, int batch, int fixed_n, int max_n, int m,
                      float* temp, int32_t* idx, int add_offset, hipStream_t st, void* multi_scratch = nullptr, int async_mode = -1) {
  if (batch <= 0 || m <= 0) return SV_OK;
  static unsigned nonce = 0;
  static const bool multi_off = getenv("SEEVCN_FPS_MULTI") && atoi(getenv("SEEVCN_FPS_MULTI")) == 0;
  // several workgroups per scene when the scene is large enough to keep them busy, they all fit the chip at once and m fits the 16-bit tag
  if (multi_scratch && !multi_off && batch * FPS_W <= 256 && max_n >= 4096 && max_n <= 8 * FPS_W * FPS_THREADS && m < 65536) {
    FpsRecord* rec = reinterpret_cast<FpsRecord*>(multi_scratch);
    int32_t* err = reinterpret_cast<int32_t*>(rec + (size_t)batch * 2 * FPS_W);
    static const int first_mode = getenv("SEEVCN_FPS_MULTI") && atoi(getenv("SEEVCN_FPS_MULTI")) == 1 ? 0 : 1;
    const int p = (max_n + FPS_W * FPS_THREADS - 1) / (FPS_W * FPS_THREADS);
    // first with the scene's workgroups on one XCD and its L2 as the meeting point (1.9 us per round); were the dispatch order ever not
    // "workgroup i -> XCD i % 8" a partner would poll a stale line of its own L2, time out and raise the error word: then once more with
    // write-through records, which hold for any placement (2.8 us per round).  The error word is read back, so this path ends synchronised.
    for (int one_xcd = async_mode < 0 ? first_mode : 1 - async_mode; one_xcd >= 0; --one_xcd) {
      SV_HIP(hipMemsetAsync(err, 0, 4, st));
      nonce = (nonce + 1) & 0xffffu;
      if (nonce == 0) nonce = 1;
      const int spin_limit = one_xcd ? 1 << 17 : 1 << 21;
      dim3 grid(one_xcd ? (batch + 7) / 8 * 8 * FPS_W : batch * FPS_W), block(FPS_THREADS);
      if (p <= 1) hipLaunchKernelGGL(k_fps_multi<1>, grid, block, 0, st, xyz, starts, counts, fixed_n, m, idx, add_offset, rec, nonce, err, batch, one_xcd, spin_limit);
      else if (p <= 2) hipLaunchKernelGGL(k_fps_multi<2>, grid, block, 0, st, xyz, starts, counts, fixed_n, m, idx, add_offset, rec, nonce, err, batch, one_xcd, spin_limit);
      else if (p <= 4) hipLaunchKernelGGL(k_fps_multi<4>, grid, block, 0, st, xyz, starts, counts, fixed_n, m, idx, add_offset, rec, nonce, err, batch, one_xcd, spin_limit);
      else hipLaunchKernelGGL(k_fps_multi<8>, grid, block, 0, st, xyz, starts, counts, fixed_n, m, idx, add_offset, rec, nonce, err, batch, one_xcd, spin_limit);
      SV_LAUNCH_CHECK();
      if (async_mode >= 0) return SV_OK;
      int32_t host_err = 0;
      SV_HIP(hipMemcpyAsync(&host_err, err, 4, hipMemcpyDeviceToHost, st));
      SV_HIP(hipStreamSynchronize(st));
      if (host_err == 0) return SV_OK;
    }
    sv_set_error("farthest_point_sampling: a partner workgroup never arrived (GPU shared with another process?); SEEVCN_FPS_MULTI=0 selects one workgroup per scene");
    return SV_ERR_HIP;
  }
  if (async_mode >= 0 && multi_scratch)      // the single-workgroup kernels cannot fail: the error word the caller will read is 0
    SV_HIP(hipMemsetAsync(reinterpret_cast<char*>(multi_scratch) + (size_t)batch * 2 * FPS_W * sizeof(FpsRecord), 0, 4, st));
  const int p = (max_n + FPS_THREADS - 1) / FPS_THREADS;
  dim3 grid(batch), block(FPS_THREADS);
  if (p <= 4) hipLaunchKernelGGL(k_fps_reg<4>, grid, block, 0, st, xyz, starts, counts, fixed_n, m, idx, add_offset);
  else if (p <= 8) hipLaunchKernelGGL(k_fps_reg<8>, grid, block, 0, st, xyz, starts, counts, fixed_n, m, idx, add_offset);
  else if (p <= 16) hipLaunchKernelGGL(k_fps_reg<16>, grid, block, 0, st, xyz, starts, counts, fixed_n, m, idx, add_offset);
  else if (p <= 32) hipLaunchKernelGGL(k_fps_reg<32>, grid, block, 0, st, xyz, starts, counts, fixed_n, m, idx, add_offset);
  else if (p <= 40) hipLaunchKernelGGL(k_fps_reg<40>, grid, block, 0, st, xyz, starts, counts, fixed_n, m, idx, add_offset);
  else if (p <= FPS_MAXP) hipLaunchKernelGGL(k_fps_reg<FPS_MAXP>, grid, block, 0, st, xyz, starts, counts, fixed_n, m, idx, add_offset);
  else {
    SV_CHECK_ARG(temp, "farthest_point_sampling: scenes with more than %d points need the temp buffer", FPS_MAXP * FPS_THREADS);
    hipLaunchKernelGGL(k_fps_stream, grid, block, 0, st, xyz, starts, counts, fixed_n, m, temp, idx, add_offset);
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// scratch of the multi-workgroup path: 2 x FPS_W records per scene + an error word (read it back to detect a missing partner; 0 = fine)
extern "C" size_t sv_fps_multi_scratch_bytes(int batch) { return (size_t)(batch > 0 ? batch : 0) * 2 * FPS_W * sizeof(FpsRecord) + 64; }

extern "C" size_t sv_fps_multi_error_offset(int batch) { return (size_t)(batch > 0 ? batch : 0) * 2 * FPS_W * sizeof(FpsRecord); }

extern "C" int sv_farthest_point_sampling(const float* xyz, int b, int n, int m, float* temp, int32_t* idx, void* stream) {
  SV_CHECK_ARG(b >= 0 && n > 0 && m >= 0 && (b == 0 || (xyz && idx)), "farthest_point_sampling: bad arguments");
  SV_CHECK_ARG((int64_t)n < (1ll << 30), "farthest_point_sampling: n too large");
  return fps_launch(xyz, nullptr, nullptr, b, n, n, m, temp, idx, 0, sv_stream(stream));
}

extern "C" int sv_stack_farthest_point_sampling(const float* xyz, const int32_t* xyz_batch_start, const int32_t* xyz_batch_cnt, int batch,
                                                int max_n, int m, float* temp, int32_t* idx, void* stream) {
  SV_CHECK_ARG(batch >= 0 && m >= 0 && max_n >= 0, "stack_farthest_point_sampling: bad arguments");
  if (batch == 0 || m == 0) return SV_OK;
  SV_CHECK_ARG(xyz && xyz_batch_start && xyz_batch_cnt && idx, "stack_farthest_point_sampling: null pointer");
  return fps_launch(xyz, xyz_batch_start, xyz_batch_cnt, batch, 0, max_n, m, temp, idx, 1, sv_stream(stream));
}

extern "C" int sv_stack_farthest_point_sampling_multi(const float* xyz, const int32_t* xyz_batch_start, const int32_t* xyz_batch_cnt, int batch,
                                                      int max_n, int m, float* temp, void* multi_scratch, int32_t* idx, void* stream) {
  SV_CHECK_ARG(batch >= 0 && m >= 0 && max_n >= 0, "stack_farthest_point_sampling: bad arguments");
  if (batch == 0 || m == 0) return SV_OK;
  SV_CHECK_ARG(xyz && xyz_batch_start && xyz_batch_cnt && idx, "stack_farthest_point_sampling: null pointer");
  return fps_launch(xyz, xyz_batch_start, xyz_batch_cnt, batch, 0, max_n, m, temp, idx, 1, sv_stream(stream), multi_scratch);
}

extern "C" int sv_stack_farthest_point_sampling_multi_async(const float* xyz, const int32_t* xyz_batch_start, const int32_t* xyz_batch_cnt, int batch,
                                                            int max_n, int m, float* temp, void* multi_scratch, int32_t* idx, int write_through,
                                                            void* stream) {
  SV_CHECK_ARG(batch >= 0 && m >= 0 && max_n >= 0, "stack_farthest_point_sampling: bad arguments");
  if (batch == 0 || m == 0) return SV_OK;
  SV_CHECK_ARG(xyz && xyz_batch_start && xyz_batch_cnt && idx && multi_scratch, "stack_farthest_point_sampling: null pointer");
  return fps_launch(xyz, xyz_batch_start, xyz_batch_cnt, batch, 0, max_n, m, temp, idx, 1, sv_stream(stream), multi_scratch, write_through ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// Ball query (stacked batches).  idx (M, nsample): the first nsample points of the query's scene, in index order,
// with |q - p|^2 < r^2; short lists are padded with the first hit; idx[0] = -1 for an empty ball
// (ball_query_gpu.cu:47-65).  One wave per query at a time; a workgroup (4 waves) walks BQ_QPB queries over the
// same scene so that the candidate tile staged in LDS is shared.
// ------------------------------------------------------------------------------------------------
constexpr int BQ_TILE = 1024;      // candidate points per LDS tile (12 KB)
constexpr int BQ_QPW = 8;          // queries per wave
constexpr int BQ_QPB = BQ_QPW * 4; // queries per workgroup

__global__ __launch_bounds__(256) void k_ball_query(int M, float radius2, int nsample, const float* __restrict__ new_xyz,
                                                    const int32_t* __restrict__ q_start, const int32_t* __restrict__ q_cnt,
                                                    const float* __restrict__ xyz, const int32_t* __restrict__ p_start,
                                                    const int32_t* __restrict__ p_cnt, int batch, int32_t* __restrict__ idx) {
  __shared__ float tile[BQ_TILE * 3];
  __shared__ int done_cnt;
  // blockIdx.y = scene, blockIdx.x = query block inside the scene
  const int b = blockIdx.y;
  const int nq = q_cnt[b], qs = q_start[b];
  const int q0 = blockIdx.x * BQ_QPB;
  if (q0 >= nq) return;
  const int np = p_cnt[b], ps = p_start[b];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float qx[BQ_QPW], qy[BQ_QPW], qz[BQ_QPW];
  int cnt[BQ_QPW];
  bool live[BQ_QPW];
#pragma unroll
  for (int u = 0; u < BQ_QPW; ++u) {
    const int q = q0 + wid * BQ_QPW + u;
    live[u] = q < nq;
    const float* p = new_xyz + (int64_t)(qs + (live[u] ? q : 0)) * 3;
    qx[u] = p[0]; qy[u] = p[1]; qz[u] = p[2];
    cnt[u] = 0;
  }
  for (int t0 = 0; t0 < np; t0 += BQ_TILE) {
    const int tn = min(BQ_TILE, np - t0);
    if (tid == 0) done_cnt = 0;
    __syncthreads();
    for (int e = tid; e < tn * 3; e += 256) tile[e] = xyz[(int64_t)(ps + t0) * 3 + e];
    __syncthreads();
    int wave_done = 1;
#pragma unroll
    for (int u = 0; u < BQ_QPW; ++u) {
      if (!live[u] || cnt[u] >= nsample) continue;
      int32_t* out = idx + (int64_t)(qs + q0 + wid * BQ_QPW + u) * nsample;
      for (int c0 = 0; c0 < tn && cnt[u] < nsample; c0 += 64) {
        const int k = c0 + lane;
        bool hit = false;
        if (k < tn) {
          const float x = tile[k * 3], y = tile[k * 3 + 1], z = tile[k * 3 + 2];
          const float d2 = (qx[u] - x) * (qx[u] - x) + (qy[u] - y) * (qy[u] - y) + (qz[u] - z) * (qz[u] - z);
          hit = d2 < radius2;
        }
        const unsigned long long m = __ballot(hit);
        if (m) {
          const int pos = cnt[u] + __popcll(m & ((1ull << lane) - 1ull));
          if (cnt[u] == 0) {  // first hit of this query: pre-fill the whole row with it (the reference's padding rule)
            const int first = t0 + c0 + (__ffsll((long long)m) - 1);
            for (int l = lane; l < nsample; l += 64) out[l] = first;
          }
          if (hit && pos < nsample) out[pos] = t0 + k;
          cnt[u] += __popcll(m);
        }
      }
      if (cnt[u] < nsample) wave_done = 0;
    }
    if (lane == 0 && wave_done) atomicAdd(&done_cnt, 1);
    __syncthreads();
    if (done_cnt == 4) break;   // every query of the workgroup is full: stop streaming candidates
  }
#pragma unroll
  for (int u = 0; u < BQ_QPW; ++u)
    if (live[u] && cnt[u] == 0 && lane == 0) idx[(int64_t)(qs + q0 + wid * BQ_QPW + u) * nsample] = -1;
}

extern "C" int sv_ball_query_stack(int batch, int M, int max_queries_per_scene, float radius, int nsample, const float* new_xyz,
                                   const int32_t* new_xyz_batch_start, const int32_t* new_xyz_batch_cnt, const float* xyz,
                                   const int32_t* xyz_batch_start, const int32_t* xyz_batch_cnt, int32_t* idx, void* stream) {
  SV_CHECK_ARG(batch >= 0 && M >= 0 && nsample > 0 && radius >= 0.f, "ball_query: bad arguments");
  if (batch == 0 || M == 0 || max_queries_per_scene == 0) return SV_OK;
  SV_CHECK_ARG(new_xyz && new_xyz_batch_start && new_xyz_batch_cnt && xyz && xyz_batch_start && xyz_batch_cnt && idx, "ball_query: null pointer");
  dim3 grid(sv_div_up(max_queries_per_scene, BQ_QPB), batch);
  hipLaunchKernelGGL(k_ball_query, grid, dim3(256), 0, sv_stream(stream), M, radius * radius, nsample, new_xyz, new_xyz_batch_start,
                     new_xyz_batch_cnt, xyz, xyz_batch_start, xyz_batch_cnt, batch, idx);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// Grouping: out (M, C, nsample)[m][c][s] = features[start(scene of m) + idx[m][s]][c]   (group_points_gpu.cu:71-102)
// row_start[m] = first feature row of the scene that query m belongs to (precomputed once per batch on the host side).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_group_points(int64_t total, int C, int nsample, const float* __restrict__ features,
                                                      const int32_t* __restrict__ idx, const int32_t* __restrict__ row_start,
                                                      float* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(e % nsample);
    const int64_t t = e / nsample;
    const int c = (int)(t % C);
    const int64_t m = t / C;
    const int32_t j = idx[m * nsample + s];
    out[e] = features[((int64_t)row_start[m] + j) * C + c];
  }
}

// One thread per (query, channel): the nsample gradients of that pair are contiguous (grad_out is (M, C, nsample)), lanes run over the
// channels so the atomic adds of a wave go to consecutive addresses of one feature row, and the slots that repeat the first neighbour
// (ball_query pads short balls that way, ball_query_gpu.cu:52-58) are summed in a register first: one add per DISTINCT neighbour instead of
// one per slot.  (One thread per element with the slot index fastest -- the reference's layout, group_points_gpu.cu:27-45 -- put 32 lanes
// on the same address for every padded ball: 99 ms at 110 592 queries x 128 channels x 16 slots.)
__global__ __launch_bounds__(256) void k_group_points_grad(int64_t pairs, int C, int nsample, const float* __restrict__ grad_out,
                                                           const int32_t* __restrict__ idx, const int32_t* __restrict__ row_start,
                                                           float* __restrict__ grad_features) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < pairs; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const int64_t m = t / C;
    const int32_t* id = idx + m * nsample;
    const float* g = grad_out + t * nsample;
    const int64_t row0 = row_start[m];
    const int32_t j0 = id[0];
    float acc = g[0];
    for (int s = 1; s < nsample; ++s) {
      const int32_t j = id[s];
      const float v = g[s];
      if (j == j0) acc += v;
      else atomicAdd(&grad_features[(row0 + j) * C + c], v);
    }
    atomicAdd(&grad_features[(row0 + j0) * C + c], acc);
  }
}

extern "C" int sv_group_points_stack(int M, int C, int nsample, const float* features, const int32_t* idx, const int32_t* row_start,
                                     float* out, void* stream) {
  SV_CHECK_ARG(M >= 0 && C > 0 && nsample > 0, "group_points: bad arguments");
  if (M == 0) return SV_OK;
  SV_CHECK_ARG(features && idx && row_start && out, "group_points: null pointer");
  const int64_t total = (int64_t)M * C * nsample;
  hipLaunchKernelGGL(k_group_points, dim3(sv_grid_1d(total, 256, 256 * 16)), dim3(256), 0, sv_stream(stream), total, C, nsample, features, idx,
                     row_start, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_group_points_grad_stack(int M, int C, int N, int nsample, const float* grad_out, const int32_t* idx,
                                          const int32_t* row_start, float* grad_features, void* stream) {
  SV_CHECK_ARG(M >= 0 && C > 0 && nsample > 0 && N >= 0, "group_points_grad: bad arguments");
  hipStream_t st = sv_stream(stream);
  if (N > 0) {
    SV_CHECK_ARG(grad_features, "group_points_grad: null pointer");
    SV_HIP(hipMemsetAsync(grad_features, 0, (size_t)N * C * 4, st));
  }
  if (M == 0) return SV_OK;
  SV_CHECK_ARG(grad_out && idx && row_start, "group_points_grad: null pointer");
  const int64_t pairs = (int64_t)M * C;
  hipLaunchKernelGGL(k_group_points_grad, dim3(sv_grid_1d(pairs, 256, 256 * 64)), dim3(256), 0, st, pairs, C, nsample, grad_out, idx, row_start,
                     grad_features);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// Ball query over a cell hash (same result as k_ball_query, element for element).  The reference scans every point of the scene for every query
// (ball_query_gpu.cu:47-65: O(M N)); here the support points are counting-sorted into hash buckets of cubic cells of edge r (1 + 1e-4) -- a hit
// differs from its query by less than r in every coordinate, so it lives in one of the 27 cells around the query's -- and a wave per query
// gathers the candidates of those 27 buckets, keeps the ones of the right scene and cell that pass the reference's own distance test
// ((qx-x)^2 + (qy-y)^2 + (qz-z)^2 < r^2, same operation order), and writes the nsample SMALLEST indices in ascending order: exactly "the first
// nsample hits in index order", padded with the first hit, idx[0] = -1 for an empty ball.
// Build: count (one atomic per point) -> exclusive scan over the buckets (one workgroup) -> fill (entries {x, y, z, index}, 16 B, so that a
// bucket is one contiguous read).  Query: lanes 0..26 fetch their bucket's range, the candidates are enumerated flat over the 27 ranges, hits
// are compacted into a wave-private LDS list and the nsample smallest are taken by a 64-lane bitonic sort (<= 64 hits) or by repeated
// minimum extraction (more).
// ------------------------------------------------------------------------------------------------
constexpr int BQH_LIST = 1024;         // hits kept per wave (more: the list is reduced to its nsample smallest and refilled)

__device__ __forceinline__ void bqh_cell(float x, float y, float z, float inv, int& cx, int& cy, int& cz) {
  cx = (int)floorf(x * inv), cy = (int)floorf(y * inv), cz = (int)floorf(z * inv);
}
__device__ __forceinline__ uint32_t bqh_bucket(int b, int cx, int cy, int cz, uint32_t mask) {
  return (((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u) ^ ((uint32_t)cz * 83492791u) ^ ((uint32_t)b * 2654435761u)) & mask;
}
__device__ __forceinline__ int bqh_scene(int i, const int32_t* __restrict__ start, const int32_t* __restrict__ cnt, int batch) {
  int b = 0;
  while (b + 1 < batch && i >= start[b] + cnt[b]) ++b;
  return b;
}

__global__ __launch_bounds__(256) void k_bqh_count(int N, const float* __restrict__ xyz, const int32_t* __restrict__ p_start,
                                                   const int32_t* __restrict__ p_cnt, int batch, float inv, uint32_t mask,
                                                   int32_t* __restrict__ count, int32_t* __restrict__ bucket_of) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  int cx, cy, cz;
  bqh_cell(xyz[(int64_t)i * 3], xyz[(int64_t)i * 3 + 1], xyz[(int64_t)i * 3 + 2], inv, cx, cy, cz);
  const uint32_t bk = bqh_bucket(bqh_scene(i, p_start, p_cnt, batch), cx, cy, cz, mask);
  bucket_of[i] = (int32_t)bk;
  atomicAdd(&count[bk], 1);
}

// exclusive scan of count[0..T) -> start[0..T], start[T] = total; one workgroup, tiles of 4096 buckets; cursor[] (= count's memory) is zeroed
__global__ __launch_bounds__(1024) void k_bqh_scan(int32_t* __restrict__ count, int T, int32_t* __restrict__ start) {
  __shared__ int32_t s_wave[16];
  __shared__ int32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 4096) {
    const int e = t0 + tid * 4;
    int4 v = make_int4(0, 0, 0, 0);
    if (e < T) v = *reinterpret_cast<const int4*>(count + e);           // T is a power of two >= 4096
    const int local = v.x + v.y + v.z + v.w;
    int inc = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off);
      if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wid] = inc;
    __syncthreads();
    int base = s_carry;
    for (int w = 0; w < wid; ++w) base += s_wave[w];
    const int ex = base + inc - local;
    if (e < T) {
      *reinterpret_cast<int4*>(start + e) = make_int4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
      *reinterpret_cast<int4*>(count + e) = make_int4(0, 0, 0, 0);
    }
    __syncthreads();
    if (tid == 1023) s_carry = ex + local;
    __syncthreads();
  }
  if (tid == 0) start[T] = s_carry;
}

__global__ __launch_bounds__(256) void k_bqh_fill(int N, const float* __restrict__ xyz, const int32_t* __restrict__ p_start,
                                                  const int32_t* __restrict__ p_cnt, int batch, const int32_t* __restrict__ bucket_of,
                                                  const int32_t* __restrict__ start, int32_t* __restrict__ cursor, float4* __restrict__ entries) {
  (void)p_start, (void)p_cnt, (void)batch;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int bk = bucket_of[i];
  const int pos = start[bk] + atomicAdd(&cursor[bk], 1);
  entries[pos] = make_float4(xyz[(int64_t)i * 3], xyz[(int64_t)i * 3 + 1], xyz[(int64_t)i * 3 + 2], __int_as_float(i));      // global row
}

// ascending 64-lane bitonic sort of one int per lane
__device__ __forceinline__ int bqh_sort64(int v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int o = __shfl_xor(v, j);
      const bool up = ((lane & k) == 0), lower = ((lane & j) == 0);
      v = (lower == up) ? min(v, o) : max(v, o);
    }
  return v;
}

// the smallest value of list[0..L) for the whole wave; that entry is then marked taken
__device__ __forceinline__ int bqh_take_min(int32_t* list, int L, int lane) {
  int best = 0x7fffffff, at = -1;
  for (int t = lane; t < L; t += 64) {
    const int v = list[t];
    if (v < best) best = v, at = t;
  }
  int wb = best;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wb = min(wb, __shfl_xor(wb, off));
  const unsigned long long own = __ballot(best == wb && at >= 0);
  if (own && lane == __ffsll((long long)own) - 1) list[at] = 0x7fffffff;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  return wb;
}

__global__ __launch_bounds__(256) void k_bqh_query(int M, const float* __restrict__ new_xyz, const int32_t* __restrict__ q_start,
                                                   const int32_t* __restrict__ q_cnt, const float* __restrict__ xyz,
                                                   const int32_t* __restrict__ p_start, const int32_t* __restrict__ p_cnt, int batch,
                                                   const int32_t* __restrict__ start, const float4* __restrict__ entries, float inv, uint32_t mask,
                                                   float radius2, int nsample, int32_t* __restrict__ idx) {
  __shared__ int32_t s_list[4][BQH_LIST];
  __shared__ int32_t s_pre[4][32], s_beg[4][32];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int32_t* list = s_list[wid];
  int32_t* pre = s_pre[wid];
  int32_t* beg = s_beg[wid];
  for (int q = blockIdx.x * 4 + wid; q < M; q += gridDim.x * 4) {
    const int b = bqh_scene(q, q_start, q_cnt, batch);
    const int ps = p_start[b], pn = p_cnt[b];
    const float qx = new_xyz[(int64_t)q * 3], qy = new_xyz[(int64_t)q * 3 + 1], qz = new_xyz[(int64_t)q * 3 + 2];
    int32_t* out = idx + (int64_t)q * nsample;
    int qcx, qcy, qcz;
    bqh_cell(qx, qy, qz, inv, qcx, qcy, qcz);
    // lanes 0..26: the bucket ranges of the 27 cells around the query's
    int n = 0;
    if (lane < 27) {
      const uint32_t bk = bqh_bucket(b, qcx + lane % 3 - 1, qcy + (lane / 3) % 3 - 1, qcz + lane / 9 - 1, mask);
      const int s = start[bk];
      n = start[bk + 1] - s;
      beg[lane] = s;
    }
    int inc = n;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const int o = __shfl_up(inc, off);
      if (lane >= off) inc += o;
    }
    if (lane < 27) pre[lane + 1] = inc;
    if (lane == 0) pre[0] = 0;
    const int total = __shfl(inc, 26);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    int L = 0;                                                        // hits in the list
    bool scan = false;
    for (int c0 = 0; c0 < total; c0 += 256) {                         // 256 candidates per round: four independent loads per lane
      int hit[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = c0 + u * 64 + lane;
        hit[u] = -1;
        if (i < total) {
          int lo = 0, hi = 27;                                        // the cell k with pre[k] <= i < pre[k + 1]
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= i) lo = mid; else hi = mid;
          }
          const float4 e = entries[beg[lo] + (i - pre[lo])];
          int ex, ey, ez;
          bqh_cell(e.x, e.y, e.z, inv, ex, ey, ez);
          const float d2 = (qx - e.x) * (qx - e.x) + (qy - e.y) * (qy - e.y) + (qz - e.z) * (qz - e.z);
          // an entry counts in its own cell only (a bucket can hold several cells, and two of the 27 cells can share a bucket) and in the
          // query's scene only (scenes share the coordinate frame; the scene is hashed in, but buckets collide)
          const int j = __float_as_int(e.w) - ps;
          if (j >= 0 && j < pn && ex == qcx + lo % 3 - 1 && ey == qcy + (lo / 3) % 3 - 1 && ez == qcz + lo / 9 - 1 && d2 < radius2) hit[u] = j;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned long long m = __ballot(hit[u] >= 0);
        if (m) {
          const int cntm = __popcll(m);
          if (L + cntm > BQH_LIST) {                                  // list full: keep its nsample smallest and go on (nsample <= 64)
            const int kept = min(nsample, L);
            int keep = 0x7fffffff;
            for (int r = 0; r < kept; ++r) {
              const int wb = bqh_take_min(list, L, lane);
              if (lane == r) keep = wb;
            }
            if (lane < kept) list[lane] = keep;
            L = kept;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          }
          if (hit[u] >= 0) list[L + __popcll(m & ((1ull << lane) - 1ull))] = hit[u];
          L += cntm;
        }
      }
      if (c0 == 0 && total > 1024) {
        // a crowded neighbourhood: from the hit rate of the first 256 candidates, would the reference's own scan in index order (which stops at
        // the nsample-th hit) need fewer rounds than the rest of the candidates?  Both in rounds of 256 points.
        const float est_hits = fmaxf((float)L, 0.5f) * (float)total * (1.f / 256.f);
        const float scan_rounds = (float)pn * (float)nsample / (est_hits * 256.f), hash_rounds = (float)(total - 256) * (1.f / 256.f);
        if (scan_rounds < hash_rounds) {
          scan = true;
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (scan) {
      int cnt = 0;
      for (int p0 = 0; p0 < pn && cnt < nsample; p0 += 256) {
        bool h[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = p0 + u * 64 + lane;
          h[u] = false;
          if (k < pn) {
            const float* p = xyz + (int64_t)(ps + k) * 3;
            const float x = p[0], y = p[1], z = p[2];
            h[u] = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z) < radius2;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned long long m = __ballot(h[u]);
          if (m && cnt < nsample) {
            const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
            if (cnt == 0) {                                           // first hit: the whole row is pre-filled with it (the reference's padding)
              const int first = p0 + u * 64 + (__ffsll((long long)m) - 1);
              if (lane < nsample) out[lane] = first;
            }
            if (h[u] && pos < nsample) out[pos] = p0 + u * 64 + lane;
            cnt += __popcll(m);
          }
        }
      }
      if (cnt == 0 && lane == 0) out[0] = -1;
      continue;
    }
    if (L == 0) {
      if (lane == 0) out[0] = -1;
    } else if (L <= 64) {
      const int v = bqh_sort64(lane < L ? list[lane] : 0x7fffffff, lane);
      const int first = __shfl(v, 0);
      if (lane < nsample) out[lane] = lane < L ? v : first;            // nsample <= 64
    } else {
      for (int r = 0; r < nsample; ++r) {                              // more than 64 hits: every slot has a hit of its own
        const int wb = bqh_take_min(list, L, lane);
        if (lane == 0) out[r] = wb;
      }
    }
  }
}

extern "C" size_t sv_ball_query_hash_scratch_bytes(int64_t n_points) {
  int64_t T = 4096;
  while (T < 2 * n_points) T <<= 1;
  return (size_t)(2 * T + 8) * 4 + (size_t)n_points * 4 + (size_t)n_points * 16 + 64;
}

// same arguments and result as sv_ball_query_stack + the number of support points and a scratch of sv_ball_query_hash_scratch_bytes(N) bytes
extern "C" int sv_ball_query_stack_hashed(int batch, int M, int64_t N, float radius, int nsample, const float* new_xyz,
                                          const int32_t* new_xyz_batch_start, const int32_t* new_xyz_batch_cnt, const float* xyz,
                                          const int32_t* xyz_batch_start, const int32_t* xyz_batch_cnt, void* scratch, int32_t* idx, void* stream) {
  SV_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0 && nsample > 0 && nsample <= 64 && radius > 0.f, "ball_query_hashed: bad arguments (nsample <= 64, radius > 0)");
  if (batch == 0 || M == 0) return SV_OK;
  SV_CHECK_ARG(new_xyz && new_xyz_batch_start && new_xyz_batch_cnt && xyz_batch_start && xyz_batch_cnt && idx && scratch && (xyz || N == 0),
               "ball_query_hashed: null pointer");
  SV_CHECK_ARG(N < (1ll << 30), "ball_query_hashed: too many points");
  hipStream_t st = sv_stream(stream);
  int64_t T = 4096;
  while (T < 2 * N) T <<= 1;
  int32_t* count = reinterpret_cast<int32_t*>(scratch);            // T     (bucket sizes, then the fill cursors)
  int32_t* start = count + T;                                      // T + 1 (+ padding to 8)
  int32_t* bucket_of = start + T + 8;                              // N
  float4* entries = reinterpret_cast<float4*>(reinterpret_cast<char*>(bucket_of + N) + ((16 - ((uintptr_t)(bucket_of + N) & 15)) & 15));
  const float inv = 1.0f / (radius * 1.0001f);
  const uint32_t mask = (uint32_t)(T - 1);
  SV_HIP(hipMemsetAsync(count, 0, (size_t)T * 4, st));
  if (N > 0) hipLaunchKernelGGL(k_bqh_count, dim3(sv_div_up(N, 256)), dim3(256), 0, st, (int)N, xyz, xyz_batch_start, xyz_batch_cnt, batch, inv, mask, count, bucket_of);
  hipLaunchKernelGGL(k_bqh_scan, dim3(1), dim3(1024), 0, st, count, (int)T, start);
  if (N > 0) hipLaunchKernelGGL(k_bqh_fill, dim3(sv_div_up(N, 256)), dim3(256), 0, st, (int)N, xyz, xyz_batch_start, xyz_batch_cnt, batch, bucket_of, start, count, entries);
  hipLaunchKernelGGL(k_bqh_query, dim3(sv_grid_1d(M, 4, 256 * 8)), dim3(256), 0, st, M, new_xyz, new_xyz_batch_start, new_xyz_batch_cnt, xyz,
                     xyz_batch_start, xyz_batch_cnt, batch, start, entries, inv, mask, radius * radius, nsample, idx);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
