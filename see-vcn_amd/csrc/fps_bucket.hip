// Farthest point sampling on spatial buckets: ONE workgroup per scene, no exchange between workgroups, and a round that touches only the
// points the new sample can change.
//
// Reference: detector3d/pcdet/ops/pointnet2/pointnet2_stack/src/sampling_gpu.cu:24-140 (and the batch twin pointnet2_batch/src/sampling_gpu.cu):
// every round recomputes the distance of ALL n points to the last pick and reduces an arg-max over all of them.  Index-exact here (same distance
// expression, same running minimum, same tie rule -- see pointnet2.hip), but:
//   * k_fps_bucket_sort puts a scene's points in Morton-cell order (one counting sort in LDS, bits dealt to the axes by extent), so that 64
//     consecutive points -- one register slot of one wave, a "bucket" -- are neighbours in space.  The ORDER is free: the samples are
//     defined on original indices (kept beside the points), so any permutation gives the same picks.
//   * k_fps_bucket keeps the points, their running distances and, one bucket per LANE, the bucket's bounding box and its largest running
//     distance in registers.  A round first prices every bucket at once: lb = the distance expression evaluated on the clamped offsets to the
//     box.  IEEE rounding is monotone, so lb <= the COMPUTED distance of every point in the box; lb >= the bucket's maximum means min(d, temp)
//     changes nothing there and the bucket is skipped -- exactly, no margin.  After a few dozen picks a new sample reaches 1-4 of a scene's ~270
//     buckets.  Only those are updated (one slot each: 64 points, one DPP max to refresh the bucket's maximum), the wave's best candidate is
//     re-read from the per-bucket maxima (one DPP max over lanes), and a wave that was not touched republishes what it had.
//   * one barrier per round, 12 candidates in LDS, every wave reduces them redundantly.  Equal distances (duplicated points, m > distinct points)
//     take a slower branch that builds the reference's tie keys from the original indices.
// Rounds: 1.9 us (16 workgroups per scene exchanging records through L2, k_fps_multi) -> see profiles/r05_*_fps_micro.txt.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int FB_THREADS = 768;             // 12 waves, 3 per SIMD: 168 VGPRs per lane -- 4 x 32 slots and the rest
constexpr int FB_NW = FB_THREADS / 64;
constexpr int FB_MAXP = 32;                 // buckets per wave (one per lane) -> n <= 24576, the register path's limit
constexpr int FB_SORT_THREADS = 1024;
constexpr int FB_CELL_BITS = 14;            // 16384 Morton cells per scene: a 64 KB histogram in LDS
constexpr int FB_CELLS = 1 << FB_CELL_BITS;

__device__ __forceinline__ unsigned long long fb_key(float d, int k, int log2t) {      // pointnet2.hip fps_key
  const unsigned int lo = (unsigned int)k & ((1u << log2t) - 1u);
  const unsigned int rev = log2t ? (__brev(lo) >> (32 - log2t)) : 0u;
  const unsigned int tie = (rev << (31 - log2t)) | ((unsigned int)k >> log2t);
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0x7FFFFFFFu - tie);
}

__device__ __forceinline__ unsigned long long fb_wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(v, d, 64);
    v = o > v ? o : v;
  }
  return v;
}

// wave-wide maximum of non-negative floats (and the -1.0f "nothing here" mark) as a uniform value: their bit patterns order as signed integers,
// so every step is ONE v_max_i32 with a DPP operand (row all-reduce: quad swaps + row rotations; then row_bcast:15 / row_bcast:31) -- no LDS crossbar,
// and no copies (the compiler's own lowering of update_dpp spends v_mov + s_nop + v_mov_dpp + v_max per step; the rounds of this kernel are made of
// instruction issue slots, see the header).  s_nop 1: a DPP operand written by the VALU instruction before it needs two wait states.
__device__ __forceinline__ float fb_wave_max_nonneg(float f) {
  int v = __float_as_int(f);
  asm volatile(
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}

// the same over lanes 0..15 only (the workgroup's 12 candidates; the other lanes hold -1): two quad swaps, the mirror of each half row, the row's mirror
__device__ __forceinline__ float fb_max16_nonneg(float f) {
  int v = __float_as_int(f);
  asm volatile(
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(v, 0));
}

// A wave's register file of one quantity: P values per lane, addressed by a WAVE-UNIFORM slot number that is only known at run time (the slot of a
// flagged bucket).  Held as ONE LLVM vector value, so that `v[s]` is the hardware's indexed register access (s_set_gpr_idx_on / v_mov /
// s_set_gpr_idx_off: three instructions) instead of control flow: a switch over the register names went through hipcc's structurizer as chains of
// flag tests and register copies, and tripled the round.  32 is the widest such vector -- hence 12 waves of 32 slots, not 8 of 48.
template <int P>
struct FbFile {
  static_assert(P == 16 || P == 32, "one indexable vector");
  float __attribute__((ext_vector_type(P))) v;
  __device__ __forceinline__ float get(int s) const { return v[s]; }
  __device__ __forceinline__ void set(int s, float x) { v[s] = x; }
};

__device__ __forceinline__ float fb_min(float a, float b) {       // fminf without the canonicalising v_max the compiler puts in front of it
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ float fb_wave_min(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float fb_wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}

__device__ __forceinline__ float fb_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

#ifdef FB_STATS
__device__ unsigned long long fb_stats[24];     // rounds x waves, waves touched, buckets flagged, tie branches (wave), tie branches (final), max flagged in one wave summed over rounds
#endif

// scratch of one scene: x | y | z | original index, npad 32-bit words each (npad = 768 * P)
__device__ __forceinline__ float* fb_plane(void* scratch, int b, int npad, int which) {
  return reinterpret_cast<float*>(scratch) + ((size_t)b * 4 + which) * npad;
}

// ------------------------------------------------------------------------------------------------
// Morton-cell order of one scene: bounding box, FB_CELL_BITS bits dealt to the axes (always to the axis whose cells are longest, so cells end
// near-cubic whatever the scene's aspect: a lidar sweep gets 7 + 7 + 0), one histogram + scan + scatter in LDS.  The order inside a cell is the
// order the atomics land in -- it does not matter (see above).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FB_SORT_THREADS) void k_fps_bucket_sort(const float* __restrict__ xyz_all, const int32_t* __restrict__ starts,
                                                                     const int32_t* __restrict__ counts, int fixed_n, int npad, void* scratch) {
  __shared__ unsigned int hist[FB_CELLS];
  __shared__ float red[6][FB_SORT_THREADS / 64];
  __shared__ unsigned int wsum[FB_SORT_THREADS / 64];
  const int b = blockIdx.x;
  const int start = starts ? starts[b] : b * fixed_n;
  const int n = min(counts ? counts[b] : fixed_n, npad);
  const float* xyz = xyz_all + (int64_t)start * 3;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float* sx = fb_plane(scratch, b, npad, 0);
  float* sy = fb_plane(scratch, b, npad, 1);
  float* sz = fb_plane(scratch, b, npad, 2);
  int32_t* sk = reinterpret_cast<int32_t*>(fb_plane(scratch, b, npad, 3));
  for (int p = max(n, 0) + tid; p < npad; p += FB_SORT_THREADS) sx[p] = 0.f, sy[p] = 0.f, sz[p] = 0.f, sk[p] = -1;
  if (n <= 0) return;
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int k = tid; k < n; k += FB_SORT_THREADS) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = xyz[k * 3 + a];
      lo[a] = fminf(lo[a], v), hi[a] = fmaxf(hi[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = fb_wave_min(lo[a]), hi[a] = fb_wave_max(hi[a]);
    if (lane == 0) red[a][wid] = lo[a], red[3 + a][wid] = hi[a];
  }
  for (int c = tid; c < FB_CELLS; c += FB_SORT_THREADS) hist[c] = 0u;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = red[a][0], hi[a] = red[3 + a][0];
    for (int w = 1; w < FB_SORT_THREADS / 64; ++w) lo[a] = fminf(lo[a], red[a][w]), hi[a] = fmaxf(hi[a], red[3 + a][w]);
  }
  // deal the bits: seq[t] = axis that owns key bit t (most significant first)
  int bits[3] = {0, 0, 0};
  unsigned int seq = 0;                                           // 2 bits per key bit
  {
    float side[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    for (int t = 0; t < FB_CELL_BITS; ++t) {
      const int a = side[0] >= side[1] ? (side[0] >= side[2] ? 0 : 2) : (side[1] >= side[2] ? 1 : 2);
      seq |= (unsigned int)a << (2 * t);
      ++bits[a];
      side[a] *= 0.5f;
    }
  }
  float scale[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) scale[a] = hi[a] > lo[a] ? (float)(1 << bits[a]) / (hi[a] - lo[a]) : 0.f;
  auto cell_of = [&](int k) {
    int c[3], rem[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = (xyz[k * 3 + a] - lo[a]) * scale[a];
      c[a] = min(max((int)v, 0), (1 << bits[a]) - 1);             // a NaN coordinate lands in cell 0: only the order is at stake
      rem[a] = bits[a];
    }
    unsigned int key = 0;
    for (int t = 0; t < FB_CELL_BITS; ++t) {
      const int a = (seq >> (2 * t)) & 3;
      const int r = a == 0 ? --rem[0] : (a == 1 ? --rem[1] : --rem[2]);
      const int ca = a == 0 ? c[0] : (a == 1 ? c[1] : c[2]);
      key = (key << 1) | ((unsigned int)(ca >> r) & 1u);
    }
    return key;
  };
  for (int k = tid; k < n; k += FB_SORT_THREADS) atomicAdd(&hist[cell_of(k)], 1u);
  __syncthreads();
  // exclusive scan: 16 consecutive cells per thread
  constexpr int PER = FB_CELLS / FB_SORT_THREADS;
  unsigned int local[PER], sum = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) local[i] = hist[tid * PER + i], sum += local[i];
  unsigned int inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  unsigned int base = inc - sum;
  for (int w = 0; w < wid; ++w) base += wsum[w];
#pragma unroll
  for (int i = 0; i < PER; ++i) hist[tid * PER + i] = base, base += local[i];
  __syncthreads();
  for (int k = tid; k < n; k += FB_SORT_THREADS) {
    const unsigned int p = atomicAdd(&hist[cell_of(k)], 1u);
    sx[p] = xyz[k * 3], sy[p] = xyz[k * 3 + 1], sz[p] = xyz[k * 3 + 2], sk[p] = k;
  }
}

// ------------------------------------------------------------------------------------------------
// The sampling itself.  Wave w, slot i holds bucket i * 12 + w (sorted positions 64 * bucket ...): neighbouring buckets sit in different waves, so
// the handful a sample reaches are updated side by side.
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(FB_THREADS) void k_fps_bucket(const float* __restrict__ xyz_all, const int32_t* __restrict__ starts,
                                                           const int32_t* __restrict__ counts, int fixed_n, int m, int npad, void* scratch,
                                                           int32_t* __restrict__ idx_all, int add_offset) {
  __shared__ unsigned short s_orig[P * FB_THREADS];               // sorted position -> original index (n <= 24576)
  __shared__ __attribute__((aligned(16))) float s_ent[2][FB_NW + 1][4];   // per wave: best distance, x, y, z; slot 12: distance -1, what lanes 12.. read
  __shared__ int s_pos[2][FB_NW + 1];
  const int b = blockIdx.x;
  const int start = starts ? starts[b] : b * fixed_n;
  const int n = counts ? counts[b] : fixed_n;
  int32_t* idx = idx_all + (int64_t)b * m;
  if (n <= 0 || m <= 0) return;
  const float* xyz = xyz_all + (int64_t)start * 3;
  int log2t = 0;
  while ((2 << log2t) <= n && log2t < 10) ++log2t;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wid in a scalar register: `wid == 0` is a scalar branch
  const float* sx = fb_plane(scratch, b, npad, 0);
  const float* sy = fb_plane(scratch, b, npad, 1);
  const float* sz = fb_plane(scratch, b, npad, 2);
  const int32_t* sk = reinterpret_cast<const int32_t*>(fb_plane(scratch, b, npad, 3));
  for (int p = tid; p < P * FB_THREADS; p += FB_THREADS) s_orig[p] = p < npad ? (unsigned short)sk[p] : (unsigned short)0;
  FbFile<P> px, py, pz, pt;
  float lox = 0.f, loy = 0.f, loz = 0.f, hix = 0.f, hiy = 0.f, hiz = 0.f, bmax = -1.f;      // of bucket (slot) `lane`
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int p = (i * FB_NW + wid) * 64 + lane;
    const bool ok = p < npad && sk[p] >= 0;
    const float x = p < npad ? sx[p] : 0.f, y = p < npad ? sy[p] : 0.f, z = p < npad ? sz[p] : 0.f, t = ok ? 1e10f : -1.f;
    px.set(i, x), py.set(i, y), pz.set(i, z), pt.set(i, t);
    const float ax = fb_wave_min(ok ? x : 3.0e38f), bx = fb_wave_max(ok ? x : -3.0e38f);
    const float ay = fb_wave_min(ok ? y : 3.0e38f), by = fb_wave_max(ok ? y : -3.0e38f);
    const float az = fb_wave_min(ok ? z : 3.0e38f), bz = fb_wave_max(ok ? z : -3.0e38f);
    const float any = fb_wave_max(t);
    if (lane == i) lox = ax, hix = bx, loy = ay, hiy = by, loz = az, hiz = bz, bmax = any;
  }
  if (tid == 0) idx[0] = add_offset ? start : 0;
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];                    // first pick is index 0 (sampling_gpu.cu:44-46)
  float wM = -1.f, wx = 0.f, wy = 0.f, wz = 0.f;                  // this wave's candidate (uniform)
  int wpos = 0;
  const unsigned long long nonempty = __ballot(bmax >= 0.f);
  const int rd = lane < FB_NW ? lane : FB_NW;                     // lanes 12.. read the slot that always says "nothing"
  if (tid == 0) s_ent[0][FB_NW][0] = -1.f, s_ent[1][FB_NW][0] = -1.f;
  __syncthreads();
#ifdef FB_STATS
  unsigned long long tacc[2][7] = {{0}};
#define FB_STAMP(k) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[touched_][k] += now_ - t_; t_ = now_; }
#else
#define FB_STAMP(k)
#endif
  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
#ifdef FB_STATS
    unsigned long long t_ = __builtin_readcyclecounter();
    int touched_ = 0;
#endif
    // 1. which of this wave's buckets can the new sample change?  Same expression, same association as the distance itself, on the clamped offsets.
    const float ex = fmaxf(fmaxf(lox - x1, x1 - hix), 0.f);
    const float ey = fmaxf(fmaxf(loy - y1, y1 - hiy), 0.f);
    const float ez = fmaxf(fmaxf(loz - z1, z1 - hiz), 0.f);
    const float lb = ex * ex + ey * ey + ez * ez;
    // round 1 takes every bucket that holds points: the candidates below must exist even where lb >= the 1e10 the distances start from
    unsigned long long mask = __ballot(lb < bmax) | (j == 1 ? nonempty : 0ull);
#ifdef FB_STATS
    touched_ = mask != 0ull;
    if (lane == 0 && j >= 64) {
      atomicAdd(&fb_stats[0], 1ull);
      if (mask) atomicAdd(&fb_stats[1], 1ull);
      atomicAdd(&fb_stats[2], (unsigned long long)__popcll(mask));
      atomicMax(&fb_stats[5 + 0], (unsigned long long)__popcll(mask));
    }
#endif
    FB_STAMP(0)
    if (mask != 0ull) {
      do {
        const int s = __builtin_ctzll(mask);
        mask &= mask - 1ull;
        const float qx = px.get(s), qy = py.get(s), qz = pz.get(s);
        const float d = (qx - x1) * (qx - x1) + (qy - y1) * (qy - y1) + (qz - z1) * (qz - z1);
        const float d2 = fb_min(d, pt.get(s));
        pt.set(s, d2);
        const float mx = fb_wave_max_nonneg(d2);
        bmax = lane == s ? mx : bmax;
      } while (mask != 0ull);
      FB_STAMP(1)
      // 2. the wave's candidate from the bucket maxima
      const float M = fb_wave_max_nonneg(bmax);
      wM = M;
      if (M >= 0.f) {
        const unsigned long long cand = __ballot(bmax == M);
        const int s = __builtin_ctzll(cand);
        const unsigned long long hm = __ballot(pt.get(s) == M);
        const bool single = __popcll(cand) == 1 && __popcll(hm) == 1;
#ifdef FB_STATS
        if (lane == 0 && !single && j >= 64) atomicAdd(&fb_stats[3], 1ull);
#endif
        if (single) {
          const int l = __builtin_ctzll(hm);
          wx = fb_readlane(px.get(s), l), wy = fb_readlane(py.get(s), l), wz = fb_readlane(pz.get(s), l);
          wpos = (s * FB_NW + wid) * 64 + l;
        } else {                                                // equal distances: the reference's tie rule on original indices
          unsigned long long best = 0ull;
          float bx = 0.f, by = 0.f, bz = 0.f;
          int bp = 0;
#pragma nounroll
          for (int r = 0; r < P; ++r) {
            if (pt.get(r) == M) {
              const int p = (r * FB_NW + wid) * 64 + lane;
              const unsigned long long key = fb_key(M, (int)s_orig[p], log2t);
              if (key > best) best = key, bx = px.get(r), by = py.get(r), bz = pz.get(r), bp = p;
            }
          }
          const unsigned long long wbest = fb_wave_max_u64(best);
          const int l = __ffsll((long long)__ballot(best == wbest)) - 1;      // keys are unique per point
          wx = fb_readlane(bx, l), wy = fb_readlane(by, l), wz = fb_readlane(bz, l);
          wpos = __builtin_amdgcn_readlane(bp, l);
        }
      }
    }
    FB_STAMP(2)
    if (lane == 0) {
      *reinterpret_cast<float4*>(s_ent[par][wid]) = make_float4(wM, wx, wy, wz);
      s_pos[par][wid] = wpos;
    }
    FB_STAMP(3)
    __syncthreads();                                            // the only barrier of the round (slots alternate by parity)
    FB_STAMP(4)
    // 3. every wave reduces the 12 candidates
    const float4 e = *reinterpret_cast<const float4*>(s_ent[par][rd]);
    const int epos = s_pos[par][rd];
    const float top = fb_max16_nonneg(e.x);
    const unsigned long long tc = __ballot(e.x == top);
    int src;
    if (__popcll(tc) == 1) {
      src = __builtin_ctzll(tc);
    } else {
      const unsigned long long key = e.x == top ? fb_key(top, (int)s_orig[epos], log2t) : 0ull;
      const unsigned long long kb = fb_wave_max_u64(key);
      src = __ffsll((long long)__ballot(key == kb)) - 1;
    }
    x1 = fb_readlane(e.y, src), y1 = fb_readlane(e.z, src), z1 = fb_readlane(e.w, src);
    if (wid == 0) {
      const int pos = __builtin_amdgcn_readlane(epos, src);       // sorted position for now
      if (lane == 0) idx[j] = pos;
    }
    FB_STAMP(5)
    FB_STAMP(6)
  }
#ifdef FB_STATS
  if (lane == 0)
    for (int t = 0; t < 2; ++t)
      for (int k = 0; k < 7; ++k) atomicAdd(&fb_stats[8 + t * 7 + k], tacc[t][k]);
#endif
  // sorted positions -> original indices (idx[1..] were written by thread 0 of this workgroup)
  __threadfence();
  __syncthreads();
  for (int j = 1 + tid; j < m; j += FB_THREADS) {
    const int p = __atomic_load_n(&idx[j], __ATOMIC_RELAXED);
    idx[j] = (add_offset ? start : 0) + (int)s_orig[p];
  }
}

template <int P>
void fb_launch(const float* xyz, const int32_t* starts, const int32_t* counts, int batch, int fixed_n, int m, int npad, void* scratch, int32_t* idx,
               int add_offset, hipStream_t st) {
  hipLaunchKernelGGL(k_fps_bucket<P>, dim3(batch), dim3(FB_THREADS), 0, st, xyz, starts, counts, fixed_n, m, npad, scratch, idx, add_offset);
}

int fb_slots(int max_n) {
  const int p = (max_n + FB_THREADS - 1) / FB_THREADS;
  return p <= 16 ? 16 : FB_MAXP;
}

}  // namespace

#ifdef FB_STATS
extern "C" int sv_fps_bucket_stats(unsigned long long* out, int reset) {
  SV_HIP(hipDeviceSynchronize());
  SV_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(fb_stats), sizeof(unsigned long long) * 24));
  if (reset) {
    unsigned long long z[24] = {0};
    SV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(fb_stats), z, sizeof(z)));
  }
  return SV_OK;
}
#endif

// SEEVCN_FPS_BUCKET=0: the callers fall back to the exhaustive kernels (A/B runs)
extern "C" int sv_fps_bucket_applies(int batch, int max_n, int m) {
  static const bool off = getenv("SEEVCN_FPS_BUCKET") && atoi(getenv("SEEVCN_FPS_BUCKET")) == 0;
  static const int min_n = getenv("SEEVCN_FPS_BUCKET_MIN") ? atoi(getenv("SEEVCN_FPS_BUCKET_MIN")) : 2048;
  return !off && batch > 0 && m > 1 && max_n >= min_n && max_n <= FB_MAXP * FB_THREADS;
}

extern "C" size_t sv_fps_bucket_scratch_bytes(int batch, int max_n) {
  if (batch <= 0 || max_n <= 0 || max_n > FB_MAXP * FB_THREADS) return 0;
  return (size_t)batch * 4 * fb_slots(max_n) * FB_THREADS * sizeof(float);
}

// xyz: stacked scenes (starts / counts per scene; idx gets GLOBAL rows) or, with starts = counts = NULL, `batch` scenes of fixed_n points each
// (idx gets scene-local rows) -- the two layouts of sv_stack_farthest_point_sampling / sv_farthest_point_sampling.
extern "C" int sv_farthest_point_sampling_bucketed(const float* xyz, const int32_t* starts, const int32_t* counts, int batch, int fixed_n, int max_n,
                                                   int m, void* scratch, int32_t* idx, void* stream) {
  SV_CHECK_ARG(batch >= 0 && m >= 0 && max_n >= 0, "farthest_point_sampling_bucketed: bad arguments");
  if (batch == 0 || m == 0) return SV_OK;
  SV_CHECK_ARG(xyz && idx && scratch && (starts != nullptr) == (counts != nullptr), "farthest_point_sampling_bucketed: null pointer");
  SV_CHECK_ARG(max_n > 0 && max_n <= FB_MAXP * FB_THREADS && (starts || fixed_n == max_n),
               "farthest_point_sampling_bucketed: scenes of at most %d points (ask sv_fps_bucket_applies first)", FB_MAXP * FB_THREADS);
  hipStream_t st = sv_stream(stream);
  const int p = fb_slots(max_n), npad = p * FB_THREADS, add = starts ? 1 : 0;
  hipLaunchKernelGGL(k_fps_bucket_sort, dim3(batch), dim3(FB_SORT_THREADS), 0, st, xyz, starts, counts, fixed_n, npad, scratch);
  SV_LAUNCH_CHECK();
  switch (p) {
    case 16: fb_launch<16>(xyz, starts, counts, batch, fixed_n, m, npad, scratch, idx, add, st); break;
    default: fb_launch<FB_MAXP>(xyz, starts, counts, batch, fixed_n, m, npad, scratch, idx, add, st); break;
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}
