// Sparse 3-D convolution: output-stationary gather-GEMM over the output-major rulebook.
//
//   Y[o][n] = epilogue( sum_k sum_c X[nbr[k][o]][c] * Wt[k][n][c] )       (rows with nbr < 0 contribute 0)
//
// One kernel serves three uses (the caller picks the table and the weight view):
//   forward        X = features,  nbr = output-major table,            Wt[k][n][c] = W[k][c_in=c][c_out=n]
//   backward-data  X = grad_out,  nbr = input-major table (nbr_in),    Wt[k][n][c] = W[k][c_in=n][c_out=c]
// and a second kernel reduces the weight gradient dW[k][c][n] = sum_o X[nbr[k][o]][c] * dY[o][n].
//
// Every output row is produced exactly once, in registers, with a fixed summation order (k ascending):
// no atomics, bitwise reproducible.  The dense per-voxel products run on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32, exact fp32) -- 16-row tiles so that a (tile, offset) pair with no neighbour is skipped.
// Algorithmic traffic per layer: 4*(N_in*C_in + N_out*C_out) + 4*K*C_in*C_out + 4*K*N_out (table) bytes.
//
// The MFMA kernel runs on a PLAN of the table (sv_conv_plan_build, once per rulebook table):
//   * rows are split into 8 contiguous REGIONS (ascending key order = scene / z-slab order), one per XCD: the workgroups of region r are
//     the ones the dispatcher places on XCD r (blockIdx % 8), so a scene's feature rows are gathered through ONE 4 MiB L2 instead of eight;
//   * inside a region rows are regrouped into 16-row tiles of equal NEIGHBOUR-MASK CLASS (counting sort, no comparison sort): a tile executes
//     an offset when any of its rows has that neighbour, so equal-mask tiles waste few MFMA steps;
//   * the regrouped table is rewritten row-major (128 B per row: 27 neighbours, mask, output row): a tile's prologue reads 2 KiB of
//     consecutive bytes instead of 27 x 16 scattered words;
//   * sv_conv_plan_tiles deals a region's tiles to its waves by descending cost in snake order (equal work per wave).
//
// Replaces the third-party spconv kernels behind SubMConv3d / SparseConv3d
// (call sites: detector3d/pcdet/models/backbones_3d/spconv_backbone.py:8-27,77-117).
#include <stdlib.h>

#include "common.h"
#include "norm.h"


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Input transform of the NEXT call (sv_conv_next_input_norm): the convolution / weight gradient reads X through y = [relu](x * scale[c] + shift[c]) --
// X is then the RAW output of the convolution below and (scale, shift) the coefficients of its BatchNorm (sv_batchnorm_finalize_forward), so that the
// normalised activations are never written to memory: the BatchNorm's elementwise pass (one read + one write of every activation tensor) disappears
// into the gathers that read the tensor anyway.  Same expression as k_bn_apply_fwd (bn_act, then fmaxf): the values a consumer sees are bit for bit
// the ones the separate pass would have stored.  Absent neighbours contribute 0, not relu(shift).
struct InNorm {
  const float* coef = nullptr;   // (2, C_in): scale | shift
  int relu = 0;
};
static thread_local InNorm g_next_in;
extern "C" int sv_conv_next_input_norm(const float* coef, int relu) {
  g_next_in.coef = coef, g_next_in.relu = relu ? 1 : 0;
  return SV_OK;
}
static InNorm take_input_norm() {
  const InNorm r = g_next_in;
  g_next_in = InNorm{};
  return r;
}

struct ConvArgs {
  const float* X;         // (n_src, Kd)
  const int32_t* nbr;     // (K, n_rows)
  const float* Wt;        // (K, Nc, Kd)
  float* Y;               // (n_rows, Nc)
  const float* bias;      // (Nc) or null        : y += bias
  const float* scale;     // (Nc) or null        : y = y*scale + shift   (folded eval-mode BatchNorm)
  const float* shift;     // (Nc) or null
  const float* residual;  // (n_rows, Nc) or null: y += residual (after scale/shift, before relu)
  int relu;
  int64_t n_rows;
  int K, Kd, Nc;
};

__device__ __forceinline__ float conv_epilogue(float v, int col, int64_t row, const ConvArgs& a) {
  if (a.bias) v += a.bias[col];
  if (a.scale) v = v * a.scale[col] + a.shift[col];
  if (a.residual) v += a.residual[row * a.Nc + col];
  if (a.relu) v = fmaxf(v, 0.f);
  return v;
}

// ------------------------------------------------------------------------------------------------
// Register-stationary, barrier-free variant (the default MFMA path).
//   * a wave owns RS_G 16-row output tiles x all Nc columns, accumulators in registers;
//   * no LDS, no workgroup barriers: both MFMA operands are loaded straight into registers — the gathered
//     source rows (A) and the 16-column weight slabs (B, shared by every wave, served by L1/L2);
//   * operands of step (k, q) + 1 are requested before step (k, q) runs on the matrix core, the neighbour
//     indices of the next active offset are fetched one offset ahead;
//   * offsets with no neighbour for any of the wave's rows are skipped entirely (64-bit activity mask from a
//     ballot pre-pass), row tiles with no neighbour skip their MFMAs;
//   * the wave's tiles are taken from RS_G distant parts of the (key-sorted) row range: neighbour density is
//     spatially correlated, striding gives every wave a mix of dense and sparse regions (measured -8 % time).
// Summation order per output element is fixed (k ascending, channels ascending) -> bitwise reproducible.
// Measured (MI355X, 64->64 submanifold layer, 134 580 rows, 1.17 M pairs): 250 us = 38 TFLOP/s algorithmic;
// matrix-core busy 38 % — waves spend their time in issue stalls, see DESIGN.md "sparse conv: what limits it".
// Two LDS-DMA restructurings (global_load_lds row gathers into a ring, offset-major weight slabs) were built, verified
// bit-compatible with the tests and measured slower (304 us with columns split over a workgroup's waves and a barrier per
// position; 266 us wave-private with the next slab prefetched) — they are in the git history (commit "Experimental
// wave-private LDS-DMA sparse conv kernel"), the findings in DESIGN.md.
// ------------------------------------------------------------------------------------------------
template <int NT, int KQ, int RS_G>
__global__ __launch_bounds__(256) void k_spconv_rs(ConvArgs a) {
  constexpr int Kd = KQ * 16, Nc = NT * 16;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const int64_t n_tiles = (a.n_rows + 15) / 16;
  const int64_t n_waves = (n_tiles + RS_G - 1) / RS_G;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wid;
  if (wave_id >= n_waves) return;
  auto tile_row0 = [&](int g) { return (wave_id + (int64_t)g * n_waves) * 16; };

  // activity mask of the kernel offsets for this wave's rows
  unsigned long long active = 0ull;
  {
    // lane -> (tile lane>>4, row lane&15); with RS_G < 4 the upper tiles alias tile 0 (harmless for an OR)
    const int64_t r = tile_row0((lane >> 4) % RS_G) + li;
    for (int k = 0; k < a.K; ++k) {
      const int32_t j = r < a.n_rows ? a.nbr[(int64_t)k * a.n_rows + r] : -1;
      if (__ballot(j >= 0)) active |= 1ull << k;
    }
  }

  f32x4 acc[RS_G][NT];
#pragma unroll
  for (int g = 0; g < RS_G; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto load_j = [&](int k, int32_t (&j)[RS_G]) {
#pragma unroll
    for (int g = 0; g < RS_G; ++g) {
      const int64_t r = tile_row0(g) + li;
      j[g] = r < a.n_rows ? a.nbr[(int64_t)k * a.n_rows + r] : -1;
    }
  };
  auto load_ab = [&](int k, int q, const int32_t (&j)[RS_G], float4 (&A)[RS_G], float4 (&B)[NT]) {
#pragma unroll
    for (int g = 0; g < RS_G; ++g)
      A[g] = j[g] >= 0 ? *reinterpret_cast<const float4*>(a.X + (int64_t)j[g] * Kd + q * 16 + kk * 4) : make_float4(0, 0, 0, 0);
    const float* w = a.Wt + ((int64_t)k * Nc + li) * Kd + q * 16 + kk * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) B[t] = *reinterpret_cast<const float4*>(w + (int64_t)t * 16 * Kd);
  };

  if (active) {
    int k = __ffsll((long long)active) - 1;
    active &= active - 1;
    int32_t jc[RS_G], jn[RS_G];
    float4 Ac[RS_G], Bc[NT], An[RS_G], Bn[NT];
    load_j(k, jc);
    load_ab(k, 0, jc, Ac, Bc);
    while (true) {
      const int kn = active ? __ffsll((long long)active) - 1 : -1;
      if (kn >= 0) load_j(kn, jn);
      bool anyg[RS_G];
#pragma unroll
      for (int g = 0; g < RS_G; ++g) anyg[g] = __ballot(jc[g] >= 0) != 0ull;
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        // request the next step's operands before running this step's MFMAs
        if (q + 1 < KQ) load_ab(k, q + 1, jc, An, Bn);
        else if (kn >= 0) load_ab(kn, 0, jn, An, Bn);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int g = 0; g < RS_G; ++g)
            if (anyg[g]) {
              acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[g].x, Bc[t].x, acc[g][t], 0, 0, 0);
              acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[g].y, Bc[t].y, acc[g][t], 0, 0, 0);
              acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[g].z, Bc[t].z, acc[g][t], 0, 0, 0);
              acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[g].w, Bc[t].w, acc[g][t], 0, 0, 0);
            }
        if (q + 1 < KQ || kn >= 0) {
#pragma unroll
          for (int g = 0; g < RS_G; ++g) Ac[g] = An[g];
#pragma unroll
          for (int t = 0; t < NT; ++t) Bc[t] = Bn[t];
        }
      }
      if (kn < 0) break;
      k = kn;
      active &= active - 1;
#pragma unroll
      for (int g = 0; g < RS_G; ++g) jc[g] = jn[g];
    }
  }
  // D layout (16x16): col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int g = 0; g < RS_G; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = tile_row0(g) + kk * 4 + r;
        const int col = t * 16 + li;
        if (row < a.n_rows) a.Y[row * Nc + col] = conv_epilogue(acc[g][t][r], col, row, a);
      }
}


template <int NT, int G>
static int launch_rs_kq(const ConvArgs& a, int kq, hipStream_t st) {
  const int64_t n_tiles = (a.n_rows + 15) / 16;
  const int64_t n_waves = (n_tiles + G - 1) / G;
  const dim3 grid((unsigned)((n_waves + 3) / 4));
  switch (kq) {
    case 1: hipLaunchKernelGGL((k_spconv_rs<NT, 1, G>), grid, dim3(256), 0, st, a); return 0;
    case 2: hipLaunchKernelGGL((k_spconv_rs<NT, 2, G>), grid, dim3(256), 0, st, a); return 0;
    case 4: hipLaunchKernelGGL((k_spconv_rs<NT, 4, G>), grid, dim3(256), 0, st, a); return 0;
    case 8: hipLaunchKernelGGL((k_spconv_rs<NT, 8, G>), grid, dim3(256), 0, st, a); return 0;
  }
  return -1;
}

static int try_launch_rs(const ConvArgs& a, hipStream_t st) {
  if (a.Kd % 16 || a.Nc % 16 || a.K > 64) return -1;
  switch (a.Nc / 16) {
    case 1: return launch_rs_kq<1, 4>(a, a.Kd / 16, st);
    case 2: return launch_rs_kq<2, 4>(a, a.Kd / 16, st);
    case 4: return launch_rs_kq<4, 4>(a, a.Kd / 16, st);
    case 8: return launch_rs_kq<8, 4>(a, a.Kd / 16, st);
  }
  return -1;
}


// ================================================================================================
// Plan of a rulebook table for the MFMA kernel
// ================================================================================================
constexpr int PL_REGIONS = 8;          // one region per XCD (MI355X: 8 XCDs, workgroup b runs on XCD b % 8)
constexpr int PL_CLASSES = 4096;       // neighbour-mask classes (class_key)
constexpr int PL_WG = 1024;            // rows per workgroup of the plan kernels; region boundaries are multiples of it (and so of 16)
constexpr int PL_ROW = 32;             // int32 per row of the regrouped table: [0..26] source rows, [27] mask, [28] output row, [29..31] unused
constexpr int RS3_KMAX = 27;

// first row of region r: regions are runs of whole PL_WG-row blocks, as equal as possible
__host__ __device__ inline int64_t plan_region_start(int64_t n_rows, int r) {
  const int64_t nblk = (n_rows + PL_WG - 1) / PL_WG;
  const int64_t s = (nblk * r / PL_REGIONS) * PL_WG;
  return s < n_rows ? s : n_rows;
}

// Mask class.  A 16-row tile executes offset k when ANY of its rows has neighbour k, so rows should share tiles with rows of (nearly) the
// same mask.  Measured on the rulebooks of the bench scenes (useful / executed MFMA steps, 8 regions): tiles of consecutive rows 0.21-0.58,
// a hash of the mask (round 1) 0.48-0.74, this key 0.72-0.87, an exact sort by mask 0.70-0.85.  The key is the MIDDLE z-plane of the mask
// (bits 9..17: the 9 in-plane neighbours, the bulk of a LiDAR surface's neighbourhood) + which of the other two planes are occupied; rows
// with an empty middle plane (the input-major table of a stride-2 conv: the mask is a function of coordinate parity) are keyed by their
// first occupied plane instead.  Equal keys -> equal in-plane pattern; the other planes only add the offsets some row actually has.
__host__ __device__ inline int class_key(unsigned m) {
  const unsigned bot = m & 0x1ffu, mid = (m >> 9) & 0x1ffu, top = (m >> 18) & 0x1ffu;
  const unsigned zs = (bot != 0u ? 1u : 0u) | (top != 0u ? 2u : 0u);
  return mid ? (int)((zs << 9) | mid) : (int)(2048u | (zs << 9) | (bot ? bot : top));
}

// "for every distinct key among the live lanes": this lane's rank inside its key group, the group's size and its first lane
__device__ __forceinline__ void wave_key_groups(int key, bool live, int& rank, int& size, int& first_lane) {
  unsigned long long todo = __ballot(live);
  const int lane = threadIdx.x & 63;
  rank = 0, size = 0, first_lane = lane;
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const int k0 = __shfl(key, first);
    const unsigned long long same = __ballot(live && key == k0);
    if (live && key == k0) {
      rank = __popcll(same & ((1ull << lane) - 1));
      size = __popcll(same);
      first_lane = first;
    }
    todo &= ~same;
  }
}

// The same three answers for keys of at most BITS bits in a FIXED number of steps: lanes with an equal key are the intersection, over the key's bits,
// of the lanes that agree with this lane on that bit (one ballot per bit) -- 12 ballots for a class key whatever the number of distinct keys among the
// 64 rows (the loop above runs once per distinct key: ~20 on consecutive rows of a LiDAR table, and the deterministic plan runs it for every row).
template <int BITS>
__device__ __forceinline__ void wave_key_groups_bits(int key, bool live, int& rank, int& size, int& first_lane) {
  const int lane = threadIdx.x & 63;
  unsigned long long same = __ballot(live);
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    const bool bit = (key >> b) & 1;
    const unsigned long long m = __ballot(bit);
    same &= bit ? m : ~m;
  }
  rank = __popcll(same & ((1ull << lane) - 1ull));
  size = __popcll(same);
  first_lane = live ? __ffsll((long long)same) - 1 : lane;
}

struct PlanArgs {
  const int32_t* masks;   // (n_rows) neighbour mask of every row (written by the rulebook builders)
  int64_t n_rows;
  int32_t* hist;          // persistent: [0 .. R*C) class counts (zero between calls), [R*C .. 2R*C) class starts, [2R*C .. 3R*C) cursors
  int32_t* perm;          // out: (n_pad) row at each position, -1 in the padding of the last tile; n_pad = 16 * ceil(n_rows / 16)
  int32_t* masks_p;       // out: (n_pad) mask of the row at each position
};

__device__ __forceinline__ int plan_region_of_row(int64_t n_rows, int64_t row) {
  int r = 0;
#pragma unroll
  for (int q = 1; q < PL_REGIONS; ++q) r += row >= plan_region_start(n_rows, q) ? 1 : 0;   // starts are non-decreasing
  return r;
}

// pass 1: per-(region, class) histogram.  One row per thread; counts go wave -> LDS -> global, so the hottest class (one mask covers
// ~20 % of the rows) sees one global atomic per 1024 rows.  A workgroup lies inside one region.
// (Tried: letting the last workgroup to arrive -- release fence + ticket -- do the scan below, and the same for the BatchNorm statistics:
// one launch less each, but 24 us instead of 7 + 6.5: every workgroup's agent-scope release writes back its XCD's L2.  A kernel boundary
// costs 1.5 us on this GPU; separate launches it is.)
__global__ __launch_bounds__(PL_WG) void k_plan_hist(PlanArgs a) {
  __shared__ int s_hist[PL_CLASSES];
  for (int i = threadIdx.x; i < PL_CLASSES; i += PL_WG) s_hist[i] = 0;
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * PL_WG + threadIdx.x;
  const bool live = row < a.n_rows;
  const unsigned m = live ? (unsigned)a.masks[row] : 0u;
  int rank, size, first_lane;
  const int key = class_key(m);
  wave_key_groups(key, live, rank, size, first_lane);
  if (live && rank == 0) atomicAdd(&s_hist[key], size);
  __syncthreads();
  const int region = plan_region_of_row(a.n_rows, (int64_t)blockIdx.x * PL_WG);
  int32_t* gh = a.hist + (size_t)region * PL_CLASSES;
  for (int i = threadIdx.x; i < PL_CLASSES; i += PL_WG)
    if (s_hist[i]) atomicAdd(&gh[i], s_hist[i]);
}

// pass 2: counts -> class starts: exclusive scan per region, counts and cursors back to zero.  One 128-thread workgroup per region, 32
// consecutive classes per thread (a single 1024-thread workgroup for all regions took 17 us: one CU moving 0.5 MB).
__global__ __launch_bounds__(128) void k_plan_scan(PlanArgs a) {
  constexpr int RC = PL_REGIONS * PL_CLASSES;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, g = r * 128 + tid;      // g: global 32-class group
  int32_t* cnt = a.hist + (size_t)g * 32;
  int v[32], sum = 0;
  {
    const i32x4* c4 = reinterpret_cast<const i32x4*>(cnt);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const i32x4 t = c4[q];
      v[4 * q] = t.x, v[4 * q + 1] = t.y, v[4 * q + 2] = t.z, v[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) sum += v[u];
  }
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __shared__ int s_wave0;
  if (tid == 63) s_wave0 = incl;
  __syncthreads();
  int run = (int)plan_region_start(a.n_rows, r) + incl - sum + (tid >= 64 ? s_wave0 : 0);
  i32x4* st4 = reinterpret_cast<i32x4*>(a.hist + RC + g * 32);
  i32x4* cu4 = reinterpret_cast<i32x4*>(a.hist + 2 * RC + g * 32);
  i32x4* cn4 = reinterpret_cast<i32x4*>(cnt);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    i32x4 t;
    t.x = run, run += v[4 * q];
    t.y = run, run += v[4 * q + 1];
    t.z = run, run += v[4 * q + 2];
    t.w = run, run += v[4 * q + 3];
    st4[q] = t;
    cu4[q] = (i32x4){0, 0, 0, 0};
    cn4[q] = (i32x4){0, 0, 0, 0};
  }
}

// pass 3: placement.  position = class start + (rows of the class placed by earlier workgroups: one global atomic per (workgroup, class))
// + (rows of the class in earlier waves of this workgroup: LDS) + rank inside the wave.  Threads past n_rows fill the padding of the last tile.
__global__ __launch_bounds__(PL_WG) void k_plan_place(PlanArgs a) {
  __shared__ int s_cnt[PL_CLASSES];
  constexpr int RC = PL_REGIONS * PL_CLASSES;
  for (int i = threadIdx.x; i < PL_CLASSES; i += PL_WG) s_cnt[i] = 0;
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * PL_WG + threadIdx.x;
  const int64_t n_pad = (a.n_rows + 15) / 16 * 16;
  const bool live = row < a.n_rows;
  const unsigned m = live ? (unsigned)a.masks[row] : 0u;
  const int key = live ? class_key(m) : 0;
  int rank, size, first_lane;
  wave_key_groups(key, live, rank, size, first_lane);
  int wave_off = 0;
  if (live && rank == 0) wave_off = atomicAdd(&s_cnt[key], size);
  wave_off = __shfl(wave_off, first_lane);
  __syncthreads();
  const int region = plan_region_of_row(a.n_rows, (int64_t)blockIdx.x * PL_WG);
  for (int i = threadIdx.x; i < PL_CLASSES; i += PL_WG) {
    const int c = s_cnt[i];
    if (c) s_cnt[i] = a.hist[RC + region * PL_CLASSES + i] + atomicAdd(&a.hist[2 * RC + region * PL_CLASSES + i], c);
  }
  __syncthreads();
  int64_t pos = -1;
  if (live) pos = (int64_t)s_cnt[key] + wave_off + rank;
  else if (row < n_pad) pos = row;
  if (pos < 0 || pos >= n_pad) return;            // the range check only matters if the persistent counters were clobbered
  a.perm[pos] = live ? (int32_t)row : -1;
  a.masks_p[pos] = (int32_t)m;
}

extern "C" size_t sv_conv_plan_persistent_bytes(void) { return (size_t)3 * PL_REGIONS * PL_CLASSES * sizeof(int32_t); }
extern "C" size_t sv_conv_plan_perm_bytes(int64_t n_rows) {
  const int64_t n_pad = ((n_rows > 0 ? n_rows : 0) + 15) / 16 * 16;
  return (size_t)(n_pad > 0 ? n_pad : 16) * sizeof(int32_t);
}

extern "C" int sv_conv_plan_build(const int32_t* masks, int64_t n_rows, void* persistent, int32_t* perm, int32_t* masks_p, void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && n_rows < (int64_t)1 << 30, "sv_conv_plan_build: 0 <= n_rows < 2^30");
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(masks && persistent && perm && masks_p, "sv_conv_plan_build: null pointer");
  PlanArgs a;
  a.masks = masks, a.n_rows = n_rows, a.hist = static_cast<int32_t*>(persistent), a.perm = perm, a.masks_p = masks_p;
  const int wgs = sv_div_up(n_rows, PL_WG);      // covers the <= 15 padding positions too: n_pad <= wgs * PL_WG
  hipStream_t st = sv_stream(stream);
  hipLaunchKernelGGL(k_plan_hist, dim3(wgs), dim3(PL_WG), 0, st, a);
  hipLaunchKernelGGL(k_plan_scan, dim3(PL_REGIONS), dim3(128), 0, st, a);
  hipLaunchKernelGGL(k_plan_place, dim3(wgs), dim3(PL_WG), 0, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// neighbour masks of a k-major table (for tables that did not come with masks from their builder)
__global__ __launch_bounds__(256) void k_row_masks(const int32_t* __restrict__ nbr, int64_t n_rows, int K, int32_t* __restrict__ masks) {
  for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n_rows; row += (int64_t)gridDim.x * 256) {
    unsigned m = 0;
    for (int k0 = 0; k0 < K; k0 += 9) {
      int32_t j[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) j[u] = k0 + u < K ? nbr[(int64_t)(k0 + u) * n_rows + row] : -1;
#pragma unroll
      for (int u = 0; u < 9; ++u) m |= j[u] >= 0 ? (1u << (k0 + u)) : 0u;
    }
    masks[row] = (int32_t)m;
  }
}
// k-major (K, n_rows) -> row-major (n_rows, 32) + masks, for tables that did not come with them from their builder
__global__ __launch_bounds__(256) void k_table_rows(const int32_t* __restrict__ nbr, int64_t n_rows, int K, int32_t* __restrict__ tab, int32_t* __restrict__ masks) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n_rows) return;
  int32_t e[PL_ROW];
#pragma unroll
  for (int k = 0; k < PL_ROW; ++k) e[k] = -1;
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < RS3_KMAX; ++k)
    if (k < K) {
      e[k] = nbr[(int64_t)k * n_rows + row];
      m |= e[k] >= 0 ? (1u << k) : 0u;
    }
  masks[row] = (int32_t)m;
  i32x4* dst = reinterpret_cast<i32x4*>(tab + row * PL_ROW);
#pragma unroll
  for (int q = 0; q < PL_ROW / 4; ++q) dst[q] = (i32x4){e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]};
}
extern "C" int sv_conv_table_rows(const int32_t* nbr, int64_t n_rows, int K, int32_t* table_rows, int32_t* masks, void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && K > 0 && K <= RS3_KMAX, "sv_conv_table_rows: 1 <= K <= %d (got %d)", RS3_KMAX, K);
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(nbr && table_rows && masks, "sv_conv_table_rows: null pointer");
  hipLaunchKernelGGL(k_table_rows, dim3(sv_div_up(n_rows, 256)), dim3(256), 0, sv_stream(stream), nbr, n_rows, K, table_rows, masks);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------ tiles -> waves
// A conv launch is ONE resident round of PL_WAVES_PER_SIMD waves on every SIMD: 8 regions x 128 workgroups of 4 waves.  Observed placement
// (tools/conv_trace.py, MAP=1; speed only, never correctness): workgroup b runs on XCD b % 8; inside an XCD the dispatcher walks the 4 shader
// engines and their CUs in turn, so workgroups j, j + 32, j + 64, j + 96 of a region stack up on one CU; the 4 waves of a workgroup go to the
// CU's 4 SIMDs in a rotation whose start varies.  A launch lasts as long as its busiest SIMD; a 16-row tile costs as many MFMA steps as it
// has kernel offsets with at least one neighbour (3 .. 27) and cannot be split.  So the deal balances CUs, and gives the 4 waves of a
// workgroup equal work (whichever SIMD each lands on): the region's tiles are counting-sorted by cost and taken in QUADS of 4 consecutive
// (near-equal) tiles; round after round the next 32 quads go to the 32 CU bins in snake order; inside a bin the rounds walk the four
// workgroups in snake order as well (one tile of the quad per wave), tile slot round / 4.  A wave works through its slots G tiles at a time
// (n_pass passes).  With >= 16 rounds (the bench's 64-channel layers have 16-34) the busiest CU carries 1.04-1.09x its XCD's mean; a
// region with a few 27-offset tiles and only ~7 rounds of 5-9-offset ones ends at up to 1.45x, because every CU gets one quad per round
// whatever it already holds.  Ranking the bins by load every round (sorted rounds) measured the same there and cost 14 us per deal
// instead of 4; a true longest-processing-time deal needs unequal tile counts per wave -- not built.
// Measured on the 64->64 layers with per-wave stamps: the round-1 snake deal of whole waves left the busiest SIMD at 1.19x (139 k rows) to
// 1.65x (66 k rows) the mean and 16 % of the SIMDs with a wave less than the others.
// Order inside a cost bucket is arbitrary: every output row is still produced by one wave with the same summation order, results do not
// depend on the deal.  One workgroup per region, everything in LDS.
#ifndef SEEVCN_PL_WAVES
#define SEEVCN_PL_WAVES 4
#endif
// 1 (round 6): on SUBMANIFOLD tables whose waves work on four tiles at a time (the 16- and 32-channel layers) those tiles are CONSECUTIVE in the
// cost-sorted list (units of G quads dealt together) instead of one tile from each of G different rounds.  A wave walks the union of its tiles' offsets and issues every tile's gather and the offset's weight loads in each step,
// whether the tile has the offset or not: with tiles of cost 27 / 12 / 8 / 5 the union is the 27 and a step carries 1.7 of 4 tiles on average
// (32 -> 32 at 250 k rows), with four tiles of one cost -- neighbours in the sorted list, mostly one mask class -- 3.0; steps per launch 82 k -> 47 k
// there (an emulation of the plan on the bench's tables); measured 68.6 -> 59 us on that layer, 23.8 -> 22.1 us at 16 -> 16.  NOT for the others: a
// wave of G costly tiles is also the launch's longest wave, and the strided tables' equal-cost tiles do not share masks -- 16 -> 32 strided 24.7 -> 28.1 us,
// 32 -> 64 strided 44 -> 50 us, 64 -> 64 on two tiles 112 -> 116 us when every table was dealt this way (profiles/r06_adj_ab.txt).  A region takes its
// table for submanifold when every row has the centre offset of a 27-offset kernel (bit 13 of every mask).  0: the round-2 deal everywhere (A/B builds).
#ifndef SEEVCN_PL_ADJ
#define SEEVCN_PL_ADJ 1
#endif
constexpr int PL_WAVES_PER_SIMD = SEEVCN_PL_WAVES;
constexpr int PL_BINS = 32;                                         // CUs per XCD
constexpr int PL_QUAD = 4;                                          // tiles dealt together: one per wave of a workgroup
constexpr int PL_REGION_WAVES = PL_BINS * PL_QUAD * PL_WAVES_PER_SIMD;   // 512 waves = 128 workgroups per region
constexpr int PL_MAX_REGION_TILES = 16384;                          // LDS bound of the deal (2 M rows per launch)
struct PlanDims {
  int32_t tile0[PL_REGIONS];    // first tile of the region
  int32_t tiles[PL_REGIONS];    // tiles of the region
  int32_t G;                    // tiles a wave works on at a time
  int32_t n_pass;               // passes: a wave has n_pass * G tile slots
};
static PlanDims plan_dims(int64_t n_rows, int G) {
  PlanDims d{};
  const int64_t n_tiles = (n_rows + 15) / 16;
  int max_tiles = 0;
  d.G = G;
  for (int r = 0; r < PL_REGIONS; ++r) {
    const int64_t s = plan_region_start(n_rows, r), e = r + 1 < PL_REGIONS ? plan_region_start(n_rows, r + 1) : n_rows;
    d.tile0[r] = (int32_t)(s / 16);
    d.tiles[r] = (int32_t)((r + 1 < PL_REGIONS ? e / 16 : n_tiles) - s / 16);
    if (d.tiles[r] > max_tiles) max_tiles = d.tiles[r];
  }
  const int quads = (max_tiles + PL_QUAD - 1) / PL_QUAD;
  // a region deals its tiles one quad per (bin, round) or -- submanifold tables on four tiles per wave, see plan_deal_quads -- in units of G consecutive
  // quads; the slot count covers both
  const int rounds = (quads + PL_BINS - 1) / PL_BINS;                            // quads per CU bin
  const int slots = (rounds + PL_WAVES_PER_SIMD - 1) / PL_WAVES_PER_SIMD;        // tiles per wave
  d.n_pass = slots > 0 ? (slots + G - 1) / G : 1;
  const int units = (quads + G - 1) / G;
  const int urounds = (units + PL_BINS - 1) / PL_BINS;
  const int upass = (urounds + PL_WAVES_PER_SIMD - 1) / PL_WAVES_PER_SIMD;
  if (SEEVCN_PL_ADJ && upass > d.n_pass) d.n_pass = upass;
  return d;
}

// The deal of a region's sorted tiles (descending cost) to its waves; called by every thread of the plan workgroup behind a barrier.
// SEEVCN_PL_LPT = 0: quads to the 32 CU bins in plain snake order (rounds 1-5).
// SEEVCN_PL_LPT = 1 (round 6): every round of 32 quads goes to the bins in order of the load they already hold -- the lightest bin takes the round's
// costliest quad (longest-processing-time dealing under "one quad per bin and round", which the slot layout needs).  The costs are skewed (a few
// 27-offset tiles, many of 5-9): snake order gives the bin of rank b the ranks b, 63 - b, 64 + b, ... whatever they cost, and the busiest CU carried
// 1.14x (139 k rows), 1.26x (66 k rows), 1.41x (strided 64 -> 64) the mean of the launch (per-wave stamps, profiles/r06_conv_trace_raw.txt; an
// emulation of the plan on the same tables reproduces 1.138 / 1.251 / 1.395 and gives 1.08 / 1.115 / 1.29 for this rule).  One wave does it: bins in
// lanes 0..31, a round = 32 readlanes to rank the loads + the slot writes.  Deterministic (ties by bin index): a table still has one plan.
#ifndef SEEVCN_PL_LPT
#define SEEVCN_PL_LPT 1
#endif
// AND of a region's masks: lanes hand in the AND of their rows' masks (all ones without a row), one LDS atomic per wave
__device__ __forceinline__ void plan_and_masks(unsigned* s_and, unsigned mine) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine &= (unsigned)__shfl_xor((int)mine, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAnd(s_and, mine);
}
// ... and only with at least four rounds of units to deal (a bin takes one unit per round whatever it costs: with two or three rounds a unit of four
// 27-offset tiles leaves its CU at 2-3x the mean; at the bench's four rounds the busiest CU of a region carries 1.0-1.2x (32 -> 32) / 1.4-1.8x (16 -> 16) the mean
// and the launches are still 14 % / 7 % shorter -- these layers are bound by their steps, not by the matrix pipe)
__device__ __forceinline__ bool plan_adjacent(int G, unsigned and_all, int nt) {
  const int units = ((nt + PL_QUAD - 1) / PL_QUAD + G - 1) / G;
  return G == 4 && ((and_all >> 13) & 1u) && (units + PL_BINS - 1) / PL_BINS >= 4;
}

template <typename CostOf>
__device__ __forceinline__ void plan_deal_quads(const uint16_t* s_sorted, CostOf cost_of, int nt, int tile0, int slots, int G, bool adjacent, int32_t* __restrict__ out, uint8_t* s_bin) {
  // s_bin: one byte of LDS per unit (the caller's: a table that is dead by now) -- the unit's cost, then its bin
  const int tid = threadIdx.x;
  const int nq = (nt + PL_QUAD - 1) / PL_QUAD;
  const int UG = (SEEVCN_PL_ADJ && adjacent) ? G : 1;               // quads per unit
  const int nu = (nq + UG - 1) / UG;
  // unit u of round j = u / 32 goes to `bin`; inside a bin the rounds walk its workgroups in snake order; the unit's quads fill the G slots of one pass
  // (not adjacent: a unit is one quad and a round fills one SLOT of the bin's workgroups, as in rounds 2-5)
  auto put = [&](int u, int j, int bin) {
    const int jm = j % PL_WAVES_PER_SIMD, wg = ((j / PL_WAVES_PER_SIMD) & 1) ? PL_WAVES_PER_SIMD - 1 - jm : jm;
    const int slot0 = (j / PL_WAVES_PER_SIMD) * UG;
    for (int g = 0; g < UG; ++g) {
      const int qd = u * UG + g;
#pragma unroll
      for (int part = 0; part < PL_QUAD; ++part) {
        const int p = qd * PL_QUAD + part;
        if (p < nt) out[(int64_t)((bin + PL_BINS * wg) * 4 + part) * slots + slot0 + g] = tile0 + s_sorted[p];
      }
    }
  };
  if constexpr (SEEVCN_PL_LPT == 0) {
    for (int u = tid; u < nu; u += 1024) {
      const int j = u / PL_BINS, pos = u % PL_BINS;                   // round of the bin, position in the round
      put(u, j, (j & 1) ? PL_BINS - 1 - pos : pos);
    }
  } else {
    for (int u = tid; u < nu; u += 1024) s_bin[u] = (uint8_t)cost_of(s_sorted[u * UG * PL_QUAD]);      // the unit's first tile is its costliest (its quads cost about the same)
    __syncthreads();
    if (tid < 64) {                                                   // the serial part: one wave, nothing but the ranking and two LDS bytes per round
      const int lane = tid;
      int key = lane;                                                 // (load << 5) | bin: unique, so a bin's rank is the number of smaller keys
      for (int j = 0; j * PL_BINS < nu; ++j) {
        int rank = 0;
#pragma unroll
        for (int o = 0; o < PL_BINS; ++o) rank += __builtin_amdgcn_readlane(key, o) < key ? 1 : 0;
        const int u = j * PL_BINS + rank;                            // the bin with the rank-th lightest load takes the round's rank-th costliest unit
        if (lane < PL_BINS && u < nu) {
          key += (int)s_bin[u] << 5;
          s_bin[u] = (uint8_t)lane;
        }
      }
    }
    __syncthreads();
    for (int u = tid; u < nu; u += 1024) put(u, u / PL_BINS, s_bin[u]);
  }
}

__global__ __launch_bounds__(1024) void k_plan_deal(const int32_t* __restrict__ masks_p, PlanDims d, int32_t* __restrict__ tile_of) {
  __shared__ uint8_t s_cost[PL_MAX_REGION_TILES];
  __shared__ uint16_t s_sorted[PL_MAX_REGION_TILES];     // tiles of the region in descending cost order
  __shared__ int s_cnt[32], s_start[32];
  __shared__ unsigned s_and_w;
  const int tid = threadIdx.x, r = blockIdx.x;
  const int nt = d.tiles[r], slots = d.n_pass * d.G;
  int32_t* out = tile_of + (int64_t)r * PL_REGION_WAVES * slots;
  for (int i = tid; i < PL_REGION_WAVES * slots; i += 1024) out[i] = -1;
  if (tid < 32) s_cnt[tid] = 0;
  if (tid == 0) s_and_w = 0xFFFFFFFFu;
  __syncthreads();
  unsigned andm = 0xFFFFFFFFu;
  for (int t = tid; t < nt; t += 1024) {
    const i32x4* mp = reinterpret_cast<const i32x4*>(masks_p + ((int64_t)d.tile0[r] + t) * 16);
    unsigned m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const i32x4 v = mp[q];
      m |= (unsigned)v.x | (unsigned)v.y | (unsigned)v.z | (unsigned)v.w;
      andm &= (v.x ? (unsigned)v.x : ~0u) & (v.y ? (unsigned)v.y : ~0u) & (v.z ? (unsigned)v.z : ~0u) & (v.w ? (unsigned)v.w : ~0u);     // padding positions carry mask 0
    }
    const int c = __popc(m) > 31 ? 31 : __popc(m);
    s_cost[t] = (uint8_t)c;
    atomicAdd(&s_cnt[c], 1);
  }
  __syncthreads();
  if (tid == 0) {                          // descending cost: the most expensive bucket first
    int acc = 0;
    for (int c = 31; c >= 0; --c) s_start[c] = acc, acc += s_cnt[c];
  }
  __syncthreads();
  if (tid < 32) s_cnt[tid] = 0;
  __syncthreads();
  for (int t = tid; t < nt; t += 1024) {
    const int c = s_cost[t];
    s_sorted[s_start[c] + atomicAdd(&s_cnt[c], 1)] = (uint16_t)t;
  }
  plan_and_masks(&s_and_w, andm);
  __syncthreads();
  const unsigned s_and = s_and_w;
  // quads of 4 consecutive tiles of the sorted list, dealt to the 32 CU bins (plan_deal_quads)
  __shared__ uint8_t s_bin[PL_MAX_REGION_TILES / PL_QUAD];
  plan_deal_quads(s_sorted, [&](int t) { return (int)s_cost[t]; }, nt, d.tile0[r], slots, d.G, plan_adjacent(d.G, s_and, nt), out, s_bin);
}

// The whole plan of a table in ONE launch: the 8 regions are independent (own classes, own positions, own tiles, own waves), so one
// 1024-thread workgroup per region runs the four passes above back to back out of LDS -- class histogram, exclusive scan, placement
// (perm / masks_p and the OR of each tile's masks), cost sort + deal -- with workgroup barriers between them instead of kernel boundaries
// and no global counters at all.  A step of the bench builds 12 plans: 12 launches instead of 48, and none of the ~5 us kernels whose
// cost is their launch.  Same placement rule (class start + rows of the class placed before), same deal; the order of the rows inside a
// class depends on LDS atomic order, as it depended on global atomic order before -- results do not depend on it.
// LDS: 2 x 16 KB class tables + 6 bytes per tile of the largest region.
struct PlanFusedArgs {
  const int32_t* masks;
  int64_t n_rows;
  int32_t* perm;
  int32_t* masks_p;
  int32_t* tile_of;
  PlanDims d;
  int max_tiles;          // tiles of the largest region (LDS layout)
  int debug;              // measurement only (SEEVCN_PLAN_DEBUG): 1 no histogram pass, 2 no perm / masks_p stores, 4 no deal
  int stable;             // every region has <= 65535 rows: the deterministic body (plan_region_body_stable)
};

// LDS of one plan workgroup and whether the deterministic body takes the table (sets a.stable)
static size_t plan_lds_bytes(PlanFusedArgs& a) {
  static const int force_atomic = getenv("SEEVCN_PLAN_ATOMIC") ? atoi(getenv("SEEVCN_PLAN_ATOMIC")) : 0;   // 1: the LDS-atomic placement (A/B runs, tests)
  int64_t big = 0;
  for (int r = 0; r < PL_REGIONS; ++r) {
    const int64_t s0 = plan_region_start(a.n_rows, r), s1 = r + 1 < PL_REGIONS ? plan_region_start(a.n_rows, r + 1) : a.n_rows;
    if (s1 - s0 > big) big = s1 - s0;
  }
  a.stable = (big <= 65535 && !force_atomic) ? 1 : 0;
  return (size_t)(a.stable ? 8 : 2) * PL_CLASSES * 4 + (size_t)a.max_tiles * 6;
}
static size_t plan_lds_bytes_for(const PlanFusedArgs& a) { return (size_t)(a.stable ? 8 : 2) * PL_CLASSES * 4 + (size_t)a.max_tiles * 6; }

// The deterministic body needs 128 KB + tiles of dynamic LDS: above 48 KB a kernel's limit has to be raised, PER DEVICE (the attribute belongs to the
// function's code object on the current device).  Returns false when this device cannot give the kernel that much (the caller then takes the body with
// LDS atomics); `which` = 0 k_plan_region, 1 k_plan_region_batch.
static bool plan_raise_lds(const void* fn, int which) {
  constexpr int MAX_DEV = 64;
  static signed char state[2][MAX_DEV] = {};                          // 0 unknown, 1 raised, -1 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return false;
  if (state[which][dev] == 0) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();                       // not an error of the call: the atomic body runs instead
    state[which][dev] = e == hipSuccess ? 1 : -1;
  }
  return state[which][dev] > 0;
}

__device__ __forceinline__ void plan_region_body(const PlanFusedArgs& a, const int r) {
  extern __shared__ int32_t s_dyn[];
  int32_t* s_start = s_dyn;                                   // [PL_CLASSES] counts, then class starts
  int32_t* s_cur = s_dyn + PL_CLASSES;                        // [PL_CLASSES] rows of the class placed so far
  uint32_t* s_tmask = reinterpret_cast<uint32_t*>(s_dyn + 2 * PL_CLASSES);            // [max_tiles] OR of the tile's 16 masks
  uint16_t* s_sorted = reinterpret_cast<uint16_t*>(s_tmask + a.max_tiles);            // [max_tiles] tiles in descending cost order
  __shared__ int s_wsum[16], s_cnt[32], s_cstart[32];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t row0 = plan_region_start(a.n_rows, r);
  const int64_t row1 = r + 1 < PL_REGIONS ? plan_region_start(a.n_rows, r + 1) : a.n_rows;
  const int64_t n_pad = (a.n_rows + 15) / 16 * 16;
  const int nt = a.d.tiles[r], slots = a.d.n_pass * a.d.G;
  int32_t* out = a.tile_of + (int64_t)r * PL_REGION_WAVES * slots;
  __shared__ unsigned s_and_w;
  unsigned andm = 0xFFFFFFFFu;
  if (tid == 0) s_and_w = 0xFFFFFFFFu;
  for (int i = tid; i < PL_CLASSES; i += 1024) s_start[i] = 0, s_cur[i] = 0;
  for (int i = tid; i < nt; i += 1024) s_tmask[i] = 0u;
  for (int i = tid; i < PL_REGION_WAVES * slots; i += 1024) out[i] = -1;
  if (tid < 32) s_cnt[tid] = 0;
  __syncthreads();
  // pass 1: class histogram of the region.  PLR_B masks per thread are requested before the first is used: one workgroup has ~31 rows per
  // thread and nothing else to hide the load latency behind (one load at a time: 30 us per plan, most of it waiting)
  constexpr int PLR_B = 8;
  for (int64_t base = row0; base < row1 && !(a.debug & 1); base += 1024 * PLR_B) {
    unsigned m[PLR_B];
#pragma unroll
    for (int u = 0; u < PLR_B; ++u) {
      const int64_t row = base + u * 1024 + tid;
      m[u] = row < row1 ? (unsigned)a.masks[row] : 0xFFFFFFFFu;            // bit 31 is never set in a mask: marks "no row"
    }
#pragma unroll
    for (int u = 0; u < PLR_B; ++u)
      if (m[u] != 0xFFFFFFFFu) atomicAdd(&s_start[class_key(m[u])], 1);    // LDS atomic per row: cheaper here than grouping the wave's keys first
  }
  __syncthreads();
  // pass 2: counts -> starts (4 consecutive classes per thread)
  {
    int v[4], sum = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = s_start[tid * 4 + u], sum += v[u];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wsum[wid] = incl;
    __syncthreads();
    int run = (int)row0 + incl - sum;
    for (int w = 0; w < wid; ++w) run += s_wsum[w];
#pragma unroll
    for (int u = 0; u < 4; ++u) s_start[tid * 4 + u] = run, run += v[u];
  }
  __syncthreads();
  // pass 3: placement + the OR of every tile's masks.  The last region also writes the padding of the last tile.
  const int64_t end = r + 1 < PL_REGIONS ? row1 : n_pad;
  for (int64_t base = row0; base < end; base += 1024 * PLR_B) {
    unsigned m[PLR_B];
#pragma unroll
    for (int u = 0; u < PLR_B; ++u) {
      const int64_t row = base + u * 1024 + tid;
      m[u] = row < row1 ? (unsigned)a.masks[row] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < PLR_B; ++u) {
      const int64_t row = base + u * 1024 + tid;
      const bool live = m[u] != 0xFFFFFFFFu;
      int64_t pos = -1;
      if (live) {
        const int key = class_key(m[u]);
        pos = (int64_t)s_start[key] + atomicAdd(&s_cur[key], 1);
      } else if (row < end) {
        pos = row;                                            // padding positions n_rows .. n_pad - 1
      }
      if (pos >= row0 && pos < n_pad) {
        if (!(a.debug & 2)) {
          a.perm[pos] = live ? (int32_t)row : -1;
          a.masks_p[pos] = live ? (int32_t)m[u] : 0;
        }
        if (live && m[u]) atomicOr(&s_tmask[(pos - row0) >> 4], m[u]);
      }
      if (live && m[u]) andm &= m[u];
    }
  }
  plan_and_masks(&s_and_w, andm);
  __syncthreads();
  const unsigned s_and = s_and_w;
  if (a.debug & 4) return;
  // pass 4: tiles by descending cost, quads dealt to the 32 CU bins in snake order (k_plan_deal).  Neighbouring tiles are of neighbouring
  // classes and cost about the same: a wave's 64 tiles hit 2-4 of the 32 counters, so the wave groups its keys before the LDS atomic
  for (int base = 0; base < nt; base += 1024) {
    const int t = base + tid;
    const bool live = t < nt;
    const int c = live ? min(__popc(s_tmask[t]), 31) : 0;
    int rank, size, first_lane;
    wave_key_groups(c, live, rank, size, first_lane);
    if (live && rank == 0) atomicAdd(&s_cnt[c], size);
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int c = 31; c >= 0; --c) s_cstart[c] = acc, acc += s_cnt[c];
  }
  __syncthreads();
  if (tid < 32) s_cnt[tid] = 0;
  __syncthreads();
  for (int base = 0; base < nt; base += 1024) {
    const int t = base + tid;
    const bool live = t < nt;
    const int c = live ? min(__popc(s_tmask[t]), 31) : 0;
    int rank, size, first_lane;
    wave_key_groups(c, live, rank, size, first_lane);
    int off = 0;
    if (live && rank == 0) off = atomicAdd(&s_cnt[c], size);
    off = __shfl(off, first_lane);
    if (live) s_sorted[s_cstart[c] + off + rank] = (uint16_t)t;
  }
  __syncthreads();
  plan_deal_quads(s_sorted, [&](int t) { return min(__popc(s_tmask[t]), 31); }, nt, a.d.tile0[r], slots, a.d.G, plan_adjacent(a.d.G, s_and, nt), out, reinterpret_cast<uint8_t*>(s_start));   // the class starts are dead: placement is over
}

// The same plan with a DETERMINISTIC order: inside a class the rows keep their table order, inside a cost bucket the tiles theirs, so a table has
// exactly one plan.  (With the LDS-atomic placement above the rows of a class land in arrival order; every output row is still computed by one wave
// in a fixed summation order, but the BatchNorm column sums the conv epilogue leaves per workgroup -- and with them the batch statistics, to ~1e-7
// -- depended on which rows shared a tile: two builds of the same table could flip the ReLU branch of an activation within an ulp of zero.)
// Every wave owns a contiguous run of the region's rows and counts / places them into ITS OWN 16-bit counter per class (16 waves x 4096 classes x
// 2 B = 128 KB of LDS, two waves per 32-bit word, updated with packed atomic adds that cannot carry while the region has <= 65535 rows); the
// counters turn into positions relative to the region start by one scan over (class, wave).  Regions of more than 65535 rows take the body above.
__device__ __forceinline__ void plan_region_body_stable(const PlanFusedArgs& a, const int r) {
  extern __shared__ int32_t s_dyn[];
  uint32_t* s_wc = reinterpret_cast<uint32_t*>(s_dyn);                                // [8][PL_CLASSES]: wave w -> half w & 1 of word [w >> 1][class]
  uint32_t* s_tmask = reinterpret_cast<uint32_t*>(s_dyn + 8 * PL_CLASSES);            // [max_tiles] OR of the tile's 16 masks
  uint16_t* s_sorted = reinterpret_cast<uint16_t*>(s_tmask + a.max_tiles);            // [max_tiles] tiles in descending cost order
  __shared__ int s_wsum[16], s_cstart[32];
  __shared__ int s_wcnt[16][32];                                                      // tiles of cost c owned by wave w (then: placed so far)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t row0 = plan_region_start(a.n_rows, r);
  const int64_t row1 = r + 1 < PL_REGIONS ? plan_region_start(a.n_rows, r + 1) : a.n_rows;
  const int64_t n_pad = (a.n_rows + 15) / 16 * 16;
  const int nt = a.d.tiles[r], slots = a.d.n_pass * a.d.G;
  int32_t* out = a.tile_of + (int64_t)r * PL_REGION_WAVES * slots;
  __shared__ unsigned s_and_w;
  unsigned andm = 0xFFFFFFFFu;
  if (tid == 0) s_and_w = 0xFFFFFFFFu;
  for (int i = tid; i < 8 * PL_CLASSES; i += 1024) s_wc[i] = 0u;
  for (int i = tid; i < nt; i += 1024) s_tmask[i] = 0u;
  for (int i = tid; i < PL_REGION_WAVES * slots; i += 1024) out[i] = -1;
  if (tid < 512) (&s_wcnt[0][0])[tid] = 0;
  __syncthreads();
  // this wave's rows: a contiguous run, a multiple of 64 long
  const int64_t per_wave = (((row1 - row0) + 15) / 16 + 63) / 64 * 64;
  const int64_t w0 = row0 + (int64_t)wid * per_wave, w1 = min(w0 + per_wave, row1);
  uint32_t* my_wc = s_wc + (size_t)(wid >> 1) * PL_CLASSES;
  const int sh = 16 * (wid & 1);
  constexpr int PLR_B = 8;
  // pass 1: per-(wave, class) counts
  for (int64_t base = w0; base < w1; base += 64 * PLR_B) {
    unsigned m[PLR_B];
#pragma unroll
    for (int u = 0; u < PLR_B; ++u) {
      const int64_t row = base + u * 64 + lane;
      m[u] = row < w1 ? (unsigned)a.masks[row] : 0xFFFFFFFFu;              // bit 31 is never set in a mask: marks "no row"
    }
#pragma unroll
    for (int u = 0; u < PLR_B; ++u)
      if (m[u] != 0xFFFFFFFFu) atomicAdd(&my_wc[class_key(m[u])], 1u << sh);
  }
  __syncthreads();
  // pass 2: counts -> positions relative to the region start, class-major then wave-major (4 consecutive classes per thread)
  {
    uint32_t wd[4][8];
    int tot[4], sum = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      tot[u] = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        wd[u][q] = s_wc[q * PL_CLASSES + tid * 4 + u];
        tot[u] += (int)(wd[u][q] & 0xffffu) + (int)(wd[u][q] >> 16);
      }
      sum += tot[u];
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wsum[wid] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wid; ++w) run += s_wsum[w];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int lo = (int)(wd[u][q] & 0xffffu), hi = (int)(wd[u][q] >> 16);
        s_wc[q * PL_CLASSES + tid * 4 + u] = (uint32_t)run | ((uint32_t)(run + lo) << 16);
        run += lo + hi;
      }
    }
  }
  __syncthreads();
  // pass 3: placement in table order + the OR of every tile's masks
  for (int64_t base = w0; base < w1; base += 64 * PLR_B) {
    unsigned m[PLR_B];
#pragma unroll
    for (int u = 0; u < PLR_B; ++u) {
      const int64_t row = base + u * 64 + lane;
      m[u] = row < w1 ? (unsigned)a.masks[row] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < PLR_B; ++u) {
      const int64_t row = base + u * 64 + lane;
      const bool live = m[u] != 0xFFFFFFFFu;
      const int key = live ? class_key(m[u]) : 0;
      int rank, size, first_lane;
      wave_key_groups_bits<12>(key, live, rank, size, first_lane);
      uint32_t old = 0;
      if (live && rank == 0) old = atomicAdd(&my_wc[key], (uint32_t)size << sh);
      old = (uint32_t)__shfl((int)old, first_lane);
      if (live) {
        const int64_t pos = row0 + (int64_t)((old >> sh) & 0xffffu) + rank;
        a.perm[pos] = (int32_t)row;
        a.masks_p[pos] = (int32_t)m[u];
        if (m[u]) atomicOr(&s_tmask[(pos - row0) >> 4], m[u]), andm &= m[u];
      }
    }
  }
  plan_and_masks(&s_and_w, andm);
  if (r + 1 == PL_REGIONS && a.n_rows + tid < n_pad) a.perm[a.n_rows + tid] = -1, a.masks_p[a.n_rows + tid] = 0;    // padding of the last tile
  __syncthreads();
  const unsigned s_and = s_and_w;
  // pass 4: tiles by descending cost (stable: ascending tile inside a cost), quads dealt to the 32 CU bins in snake order (k_plan_deal)
  const int tiles_per_wave = ((nt + 15) / 16 + 63) / 64 * 64;
  const int t0 = wid * tiles_per_wave, t1 = min(t0 + tiles_per_wave, nt);
  for (int base = t0; base < t1; base += 64) {
    const int t = base + lane;
    const bool live = t < t1;
    const int c = live ? min(__popc(s_tmask[t]), 31) : 0;
    int rank, size, first_lane;
    wave_key_groups_bits<5>(c, live, rank, size, first_lane);
    if (live && rank == 0) s_wcnt[wid][c] += size;                     // the wave's own row of counters: no other wave touches it
  }
  __syncthreads();
  if (tid < 32) {                                                       // cost tid: exclusive prefix over the waves; then the bucket starts, most expensive first
    int run = 0;
    for (int w = 0; w < 16; ++w) {
      const int c = s_wcnt[w][tid];
      s_wcnt[w][tid] = run;
      run += c;
    }
    s_cstart[tid] = run;                                                // total of the cost, turned into its start below
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int c = 31; c >= 0; --c) {
      const int n = s_cstart[c];
      s_cstart[c] = acc, acc += n;
    }
  }
  __syncthreads();
  for (int base = t0; base < t1; base += 64) {
    const int t = base + lane;
    const bool live = t < t1;
    const int c = live ? min(__popc(s_tmask[t]), 31) : 0;
    int rank, size, first_lane;
    wave_key_groups_bits<5>(c, live, rank, size, first_lane);
    int off = 0;
    if (live && rank == 0) off = s_wcnt[wid][c], s_wcnt[wid][c] = off + size;
    off = __shfl(off, first_lane);
    if (live) s_sorted[s_cstart[c] + off + rank] = (uint16_t)t;
  }
  __syncthreads();
  plan_deal_quads(s_sorted, [&](int t) { return min(__popc(s_tmask[t]), 31); }, nt, a.d.tile0[r], slots, a.d.G, plan_adjacent(a.d.G, s_and, nt), out, reinterpret_cast<uint8_t*>(s_wc));      // the class counters are dead: placement is over
}

__device__ __forceinline__ void plan_region_dispatch(const PlanFusedArgs& a, const int r) {
  if (a.stable) plan_region_body_stable(a, r);
  else plan_region_body(a, r);
}
__global__ __launch_bounds__(1024) void k_plan_region(PlanFusedArgs a) { plan_region_dispatch(a, blockIdx.x); }

// The plans of SEVERAL tables in one launch: workgroup b builds region b % 8 of table b / 8.  A step of the bench needs 12 plans; one
// workgroup per region and table is 96 workgroups side by side instead of 12 launches of 8 (29 us each, 8 of 256 CUs busy).
constexpr int PL_BATCH_MAX = 16;
struct PlanBatchArgs {
  PlanFusedArgs j[PL_BATCH_MAX];
};
static_assert(sizeof(PlanBatchArgs) <= 3900, "kernel argument block");
__global__ __launch_bounds__(1024) void k_plan_region_batch(PlanBatchArgs b) { plan_region_dispatch(b.j[blockIdx.x / PL_REGIONS], blockIdx.x % PL_REGIONS); }

// Tiles a wave holds in registers at a time: 2 for the 64-column kernels (113 VGPRs: four waves per SIMD), 4 for the narrow ones (their
// MFMA work per weight load is small).  The weight loads are shared by the G tiles of a pass.
static int conv_tiles_per_wave(int64_t n_rows, int Kd, int Nc) {
  (void)Kd;
  if (Nc <= 32) return 4;
  // 64-column kernels: 2 tiles per pass, but a table with no more tiles than the launch has waves (8 x 512) gives every wave ONE tile --
  // with 2 per wave half the SIMD slots stay empty and the waves that run have nobody to hide their load latency behind
  static const int64_t g1_tiles = getenv("SEEVCN_CONV_G1_TILES") ? atoll(getenv("SEEVCN_CONV_G1_TILES")) : (int64_t)PL_REGIONS * PL_REGION_WAVES * 9 / 8;
  return (n_rows + 15) / 16 <= g1_tiles ? 1 : 2;
}
extern "C" int sv_conv_tiles_per_wave(int64_t n_rows, int Kd, int Nc) { return conv_tiles_per_wave(n_rows < 0 ? 0 : n_rows, Kd, Nc); }
extern "C" size_t sv_conv_plan_tiles_bytes(int64_t n_rows, int tiles_per_wave) {
  if (tiles_per_wave < 1) tiles_per_wave = 1;
  const PlanDims d = plan_dims(n_rows < 0 ? 0 : n_rows, tiles_per_wave);
  return (size_t)PL_REGIONS * PL_REGION_WAVES * d.n_pass * d.G * sizeof(int32_t);
}

extern "C" int sv_conv_plan_tiles(const int32_t* masks_p, int64_t n_rows, int tiles_per_wave, int32_t* tile_of, void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && tiles_per_wave >= 1 && tiles_per_wave <= 4, "sv_conv_plan_tiles: bad sizes (tiles_per_wave %d)", tiles_per_wave);
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(masks_p && tile_of, "sv_conv_plan_tiles: null pointer");
  const PlanDims d = plan_dims(n_rows, tiles_per_wave);
  for (int r = 0; r < PL_REGIONS; ++r) SV_CHECK_ARG(d.tiles[r] <= PL_MAX_REGION_TILES, "sv_conv_plan_tiles: at most %d tiles per region", PL_MAX_REGION_TILES);
  hipLaunchKernelGGL(k_plan_deal, dim3(PL_REGIONS), dim3(1024), 0, sv_stream(stream), masks_p, d, tile_of);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// perm + masks_p + tile_of(tiles_per_wave) of a table in one launch (k_plan_region); same outputs as sv_conv_plan_build followed by
// sv_conv_plan_tiles up to the order of the rows inside a class
extern "C" int sv_conv_plan_build_dealt(const int32_t* masks, int64_t n_rows, int tiles_per_wave, int32_t* perm, int32_t* masks_p, int32_t* tile_of,
                                        void* stream) {
  SV_CHECK_ARG(n_rows >= 0 && n_rows < (int64_t)1 << 30 && tiles_per_wave >= 1 && tiles_per_wave <= 4, "sv_conv_plan_build_dealt: bad sizes");
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(masks && perm && masks_p && tile_of, "sv_conv_plan_build_dealt: null pointer");
  PlanFusedArgs a;
  a.masks = masks, a.n_rows = n_rows, a.perm = perm, a.masks_p = masks_p, a.tile_of = tile_of;
  a.d = plan_dims(n_rows, tiles_per_wave);
  static const int plan_debug = getenv("SEEVCN_PLAN_DEBUG") ? atoi(getenv("SEEVCN_PLAN_DEBUG")) : 0;
  a.debug = plan_debug;
  a.max_tiles = 1;
  for (int r = 0; r < PL_REGIONS; ++r) {
    SV_CHECK_ARG(a.d.tiles[r] <= PL_MAX_REGION_TILES, "sv_conv_plan_build_dealt: at most %d tiles per region", PL_MAX_REGION_TILES);
    if (a.d.tiles[r] > a.max_tiles) a.max_tiles = a.d.tiles[r];
  }
  a.max_tiles = (a.max_tiles + 1) & ~1;                                  // keeps the uint16 array 4-byte aligned
  size_t lds = plan_lds_bytes(a);
  if (lds > 48 * 1024 && !plan_raise_lds(reinterpret_cast<const void*>(k_plan_region), 0)) {
    a.stable = 0;                                                       // no large LDS on this device: the body with LDS atomics (32 KB + tiles)
    lds = plan_lds_bytes_for(a);
  }
  hipLaunchKernelGGL(k_plan_region, dim3(PL_REGIONS), dim3(1024), lds, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// jobs_host: n_jobs rows of 8 int64 = {masks, n_rows, tiles_per_wave, perm, masks_p, tile_of, 0, 0}: sv_conv_plan_build_dealt for every row, all
// in one launch (groups of PL_BATCH_MAX tables)
extern "C" int sv_conv_plan_build_dealt_batch(const int64_t* jobs_host, int n_jobs, void* stream) {
  SV_CHECK_ARG(n_jobs >= 0 && (jobs_host || n_jobs == 0), "sv_conv_plan_build_dealt_batch: bad arguments");
  hipStream_t st = sv_stream(stream);
  PlanBatchArgs b;
  int nb = 0;
  size_t lds = 0;
  auto flush = [&]() -> int {
    if (nb == 0) return SV_OK;
    if (lds > 48 * 1024 && !plan_raise_lds(reinterpret_cast<const void*>(k_plan_region_batch), 1)) {
      lds = 0;
      for (int q = 0; q < nb; ++q) {
        b.j[q].stable = 0;
        const size_t need = plan_lds_bytes_for(b.j[q]);
        if (need > lds) lds = need;
      }
    }
    hipLaunchKernelGGL(k_plan_region_batch, dim3(PL_REGIONS * nb), dim3(1024), lds, st, b);
    nb = 0, lds = 0;
    return SV_OK;
  };
  for (int q = 0; q < n_jobs; ++q) {
    const int64_t* r = jobs_host + 8 * q;
    const int64_t n_rows = r[1];
    const int g = (int)r[2];
    SV_CHECK_ARG(n_rows >= 0 && n_rows < (int64_t)1 << 30 && g >= 1 && g <= 4, "sv_conv_plan_build_dealt_batch: job %d: bad sizes", q);
    if (n_rows == 0) continue;
    SV_CHECK_ARG(r[0] && r[3] && r[4] && r[5], "sv_conv_plan_build_dealt_batch: job %d: null pointer", q);
    PlanFusedArgs& a = b.j[nb];
    a.masks = reinterpret_cast<const int32_t*>(r[0]), a.n_rows = n_rows, a.perm = reinterpret_cast<int32_t*>(r[3]);
    a.masks_p = reinterpret_cast<int32_t*>(r[4]), a.tile_of = reinterpret_cast<int32_t*>(r[5]);
    a.d = plan_dims(n_rows, g);
    a.debug = 0;
    a.max_tiles = 1;
    for (int rg = 0; rg < PL_REGIONS; ++rg) {
      SV_CHECK_ARG(a.d.tiles[rg] <= PL_MAX_REGION_TILES, "sv_conv_plan_build_dealt_batch: at most %d tiles per region", PL_MAX_REGION_TILES);
      if (a.d.tiles[rg] > a.max_tiles) a.max_tiles = a.d.tiles[rg];
    }
    a.max_tiles = (a.max_tiles + 1) & ~1;
    const size_t need = plan_lds_bytes(a);
    if (need > lds) lds = need;
    if (++nb == PL_BATCH_MAX) {
      int rc = flush();
      if (rc) return rc;
    }
  }
  int rc = flush();
  if (rc) return rc;
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------ weights in MFMA fragment order
// One contiguous KiB per (offset, 16-channel step, column tile): the B-operand load of a wave touches 8 whole cache lines instead of 16
// half lines at 16 different rows.  Both directions of a layer in one launch, into caller-owned buffers (cached by the host per weight
// version; round 1 kept ONE device-global buffer, which two streams would have raced on):
//   fwd:  Wt[k][n = c_out][c = c_in]      bwd:  Wt[k][n = c_in][c = c_out]         from W (K, C_in, C_out) given by its element strides
struct WStride {
  int64_t k, i, o;     // element strides of the (K, C_in, C_out) weight view
};
__global__ __launch_bounds__(256) void k_weight_fragments(const float* __restrict__ w, WStride ws, int K, int Cin, int Cout, float* __restrict__ wf_fwd,
                                                          float* __restrict__ wf_bwd) {
  const int total = K * Cin * Cout / 4;                      // float4 units per direction
  for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * total; i += gridDim.x * 256) {
    const bool bwd = i >= total;
    const int e = bwd ? i - total : i;
    const int Nc = bwd ? Cin : Cout, Kd = bwd ? Cout : Cin;
    const int64_t sn = bwd ? ws.i : ws.o, sc = bwd ? ws.o : ws.i;
    const int KQ = Kd / 16, NT = Nc / 16;
    const int lane = e & 63, t = (e >> 6) % NT, q = ((e >> 6) / NT) % KQ, k = (e >> 6) / (NT * KQ);
    const int li = lane & 15, kk = lane >> 4;
    const float* src = w + k * ws.k + (t * 16 + li) * sn + (q * 16 + kk * 4) * sc;
    float4 v;
    if (sc == 1 && (((uintptr_t)src) & 15) == 0) v = *reinterpret_cast<const float4*>(src);
    else v = make_float4(src[0], src[sc], src[2 * sc], src[3 * sc]);
    float* dst = bwd ? wf_bwd : wf_fwd;
    if (dst) reinterpret_cast<float4*>(dst)[e] = v;
  }
}

extern "C" int sv_conv_weight_fragments(const float* W, int64_t stride_k, int64_t stride_cin, int64_t stride_cout, int K, int Cin, int Cout, float* frag_fwd,
                                        float* frag_bwd, void* stream) {
  SV_CHECK_ARG(W && K > 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0, "sv_conv_weight_fragments: channels must be multiples of 16");
  SV_CHECK_ARG(frag_fwd || frag_bwd, "sv_conv_weight_fragments: no output");
  SV_CHECK_ARG(((uintptr_t)frag_fwd % 16 == 0) && ((uintptr_t)frag_bwd % 16 == 0), "sv_conv_weight_fragments: outputs must be 16-byte aligned");
  const WStride ws{stride_k, stride_cin, stride_cout};
  hipLaunchKernelGGL(k_weight_fragments, dim3(sv_grid_1d((int64_t)K * Cin * Cout / 2, 256)), dim3(256), 0, sv_stream(stream), W, ws, K, Cin, Cout, frag_fwd,
                     frag_bwd);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// All layers of a network in ONE launch: descs (n, 10) int64 on the device = {W, stride_k, stride_cin, stride_cout, K, C_in, C_out, frag_fwd,
// frag_bwd, first float4 unit of the layer in the launch}; a step of the bench re-lays 11 layers (the fragments follow the weights every
// forward, see spconv/functional.py), one 4 us launch each before.
struct FragDesc {
  const float* w;
  int64_t sk, si, so, K, Cin, Cout;
  float* fwd;
  float* bwd;
  int64_t unit0;
};
__global__ __launch_bounds__(256) void k_weight_fragments_batch(const FragDesc* __restrict__ descs, int n, int64_t total_units) {
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total_units; u += (int64_t)gridDim.x * 256) {
    int l = 0;
    while (l + 1 < n && descs[l + 1].unit0 <= u) ++l;
    const FragDesc d = descs[l];
    const int K = (int)d.K, Cin = (int)d.Cin, Cout = (int)d.Cout;
    const int total = K * Cin * Cout / 4;
    const int i = (int)(u - d.unit0);
    const bool bwd = i >= total;
    const int e = bwd ? i - total : i;
    const int Nc = bwd ? Cin : Cout, Kd = bwd ? Cout : Cin;
    const int64_t sn = bwd ? d.si : d.so, sc = bwd ? d.so : d.si;
    const int KQ = Kd / 16, NT = Nc / 16;
    const int lane = e & 63, t = (e >> 6) % NT, q = ((e >> 6) / NT) % KQ, k = (e >> 6) / (NT * KQ);
    const int li = lane & 15, kk = lane >> 4;
    const float* src = d.w + k * d.sk + (t * 16 + li) * sn + (q * 16 + kk * 4) * sc;
    float4 v;
    if (sc == 1 && (((uintptr_t)src) & 15) == 0) v = *reinterpret_cast<const float4*>(src);
    else v = make_float4(src[0], src[sc], src[2 * sc], src[3 * sc]);
    reinterpret_cast<float4*>(bwd ? d.bwd : d.fwd)[e] = v;
  }
}

extern "C" int sv_conv_weight_fragments_batch(const void* descs_device, int n_layers, int64_t total_units, void* stream) {
  SV_CHECK_ARG(n_layers >= 0 && total_units >= 0, "sv_conv_weight_fragments_batch: bad sizes");
  if (n_layers == 0 || total_units == 0) return SV_OK;
  SV_CHECK_ARG(descs_device, "sv_conv_weight_fragments_batch: null pointer");
  hipLaunchKernelGGL(k_weight_fragments_batch, dim3(sv_grid_1d(total_units, 256, 2048)), dim3(256), 0, sv_stream(stream),
                     static_cast<const FragDesc*>(descs_device), n_layers, total_units);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// The MFMA kernel on a plan.  Register-stationary like k_spconv_rs above, plus:
//   * PMC on k_spconv_rs: matrix core busy 38 %, waves waiting for operands that were requested only one 64-MFMA step (~2000 cycles)
//     earlier, less than the gather latency under load; hipcc additionally sinks its own prefetch loads towards their first use.  Here
//     every operand load of the loop is an inline-asm buffer_load_dwordx4 (hipcc can neither move it nor wait for it), issued TWO steps
//     ahead into a 3-deep register ring, and retired with a counted s_waitcnt vmcnt(2 x loads-per-step) that names the stage's registers
//     ("+v", form (ii) of cdna_hip_programming.md 5.7).  Every step issues exactly RS_G + NT loads: rows without a neighbour use an
//     out-of-range buffer offset (the range check returns zeros without a memory access), steps past the end issue out-of-range dummies;
//   * a wave's tiles come from the plan: tile_of[wave][slot] inside the region of its XCD (blockIdx.x % 8), rows + masks + output rows
//     from the regrouped row-major table (one 128-byte line per row);
//   * blockIdx.y selects a block of 64 output columns (C_out = 128 runs as two column blocks that gather the same rows);
//   * the wave's neighbour indices are parked in LDS once, so the loop contains no compiler-visible VMEM load.
// Same ownership, skipping and summation order (k ascending, channels ascending) as k_spconv_rs -> bitwise reproducible, and bit-identical to
// the ungrouped kernels.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ i32x4 make_srd(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));      // stride 0
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ f32x4 buf_load_b128(i32x4 srd, uint32_t voff) {
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(srd) : "memory");
  return v;
}
// address = base + soff (SGPR) + voff (VGPR) + IMM; the range check looks at voff + IMM only, so an out-of-range voff still returns zeros
// without a memory access whatever soff is
// cache-policy bits of the operand loads (measurement switches: tools/build_variant.sh x "-DSEEVCN_RS3_ROW_POLICY=1"): 0 plain (default), 1 nt
// (streaming), 2 sc1, 3 sc0 sc1
#ifndef SEEVCN_RS3_ROW_POLICY
#define SEEVCN_RS3_ROW_POLICY 0
#endif
#ifndef SEEVCN_RS3_W_POLICY
#define SEEVCN_RS3_W_POLICY 0
#endif
#if SEEVCN_RS3_ROW_POLICY == 1
#define SEEVCN_RS3_ROW_BITS " nt"
#elif SEEVCN_RS3_ROW_POLICY == 2
#define SEEVCN_RS3_ROW_BITS " sc1"
#elif SEEVCN_RS3_ROW_POLICY == 3
#define SEEVCN_RS3_ROW_BITS " sc0 sc1"
#else
#define SEEVCN_RS3_ROW_BITS ""
#endif
#if SEEVCN_RS3_W_POLICY == 1
#define SEEVCN_RS3_W_BITS " nt"
#elif SEEVCN_RS3_W_POLICY == 2
#define SEEVCN_RS3_W_BITS " sc1"
#elif SEEVCN_RS3_W_POLICY == 3
#define SEEVCN_RS3_W_BITS " sc0 sc1"
#else
#define SEEVCN_RS3_W_BITS ""
#endif
template <int IMM>
__device__ __forceinline__ f32x4 buf_load_b128_s(i32x4 srd, uint32_t voff, uint32_t soff) {
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" SEEVCN_RS3_ROW_BITS : "=v"(v) : "v"(voff), "s"(srd), "s"(soff), "i"(IMM) : "memory");
  return v;
}
template <int IMM>
__device__ __forceinline__ f32x4 buf_load_b128_w(i32x4 srd, uint32_t voff, uint32_t soff) {
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" SEEVCN_RS3_W_BITS : "=v"(v) : "v"(voff), "s"(srd), "s"(soff), "i"(IMM) : "memory");
  return v;
}

struct PlanView {
  const int32_t* tab;       // (n_rows, PL_ROW) row-major table, natural row order
  const int32_t* perm;      // (n_pad) row at each position (-1 padding)
  const int32_t* masks_p;   // (n_pad) its mask
  const int32_t* tile_of;   // [wave][G]
  PlanDims d;
  int k_flip;               // read table entry K-1-k for offset k (a submanifold table serving its own data gradient)
  int nc_total;             // columns of Y and of the weight fragments (a.Nc is the block's share)
  float* bn_partial;        // (gridDim.x, 2, nc_total) per-workgroup column sums / sums of squares of Y, or null (BatchNorm statistics made here)
  // BatchNorm BACKWARD sums instead (a data-gradient launch whose output Y is the gradient dy of a BatchNorm(+ReLU) output): bn_x = that
  // BatchNorm's input (n_rows, nc_total), its saved batch statistics and affine parameters; the partials are {sum of masked dy, sum of masked
  // dy * xhat} -- what k_bn_reduce<true> makes in a pass of its own.  bn_x null: forward statistics of Y.
  const float* bn_x;
  const float* bn_mean;
  const float* bn_istd;
  const float* bn_gamma;    // may be null (1)
  const float* bn_beta;     // may be null (0)
  int bn_relu;
  int debug;                // measurement only (SEEVCN_RS3_DEBUG; results are wrong): 1 no gathered-row loads, 2 no weight loads, 4 no MFMAs,
                            // 8 / 16 weight / row loads of a wave all at ONE address (one cache line per load instead of 16)
  unsigned long long* trace;   // measurement only (sv_debug_conv_trace): 8 words per wave, or null
  int prio;                 // SEEVCN_RS3_PRIO (A/B): 1 = s_setprio 3 for a pass's prologue, 2 = for its epilogue too; the main loop runs at 0
  const float* in_coef;     // FIN instances: (2, Kd) scale | shift applied to every gathered X value (+ ReLU when in_relu); see InNorm
  int in_relu;
};

// DBG: 0 production; 1 the measurement switches of PlanView::debug (+ trace); 2 per-wave trace only (the production loop + a few s_memtime per pass)
#ifndef SEEVCN_RS3_RA
#define SEEVCN_RS3_RA 3
#endif
#ifndef SEEVCN_RS3_RB
#define SEEVCN_RS3_RB 2
#endif
#ifndef SEEVCN_RS3_DEEP
#define SEEVCN_RS3_DEEP 0             // 1: four row and four weight stages for the one-tile 64-column instances (A/B builds; slower: see DEEP below)
#endif
#ifndef SEEVCN_RS3_OFFSET_LOOP
#define SEEVCN_RS3_OFFSET_LOOP 1      // 0: the step-by-step iterators for every instance (A/B builds)
#endif
constexpr int RS3_RA = SEEVCN_RS3_RA;      // stages of the row ring (RS3_RA - 1 steps of gathers in flight)
constexpr int RS3_RB = SEEVCN_RS3_RB;      // stages of the weight ring
// FIN: the gathered rows go through pv.in_coef (BatchNorm + ReLU of the layer below applied on load) -- production instances only
// DYN (one tile per pass, production instances): the four waves of a workgroup take the workgroup's tiles -- the union of the four waves' slots in
// the plan, heaviest level first -- one at a time from a counter in LDS instead of each walking its own slots.  A wave's length is then the
// workgroup's work / 4 up to one light tile, whatever the tiles cost one by one (dealt statically, waves of the 139 k-row layer ran 14 .. 37
// (tile, offset) steps around a mean of 21.7, and a long wave alone on its SIMD cannot fill the matrix pipe: it waits for its own gathers).  Which
// wave computes a tile does not change the tile's values; the BatchNorm column sums are kept PER LIST POSITION in LDS and combined in list order, so
// the workgroup's partial sums -- and with them the training step -- stay bit-reproducible.
constexpr int RS3_DYN_COLS = 1024;         // floats of one kind (sum / sum of squares) in the per-position array: positions x 16 NT columns
template <int NT, int KQ, int RS_G, int DBG = 0, bool FIN = false, bool DYN = false>
__global__ __launch_bounds__(256, PL_WAVES_PER_SIMD) void k_spconv_rs3(ConvArgs a, PlanView pv, const float* __restrict__ wfrag, uint32_t x_bytes,
                                                                        uint32_t w_bytes) {
  constexpr int Kd = KQ * 16;
  __shared__ int32_t s_idx_all[4][RS3_KMAX + 1][64];       // [k][lane]: source row of (tile lane>>4, row lane&15); [27][lane]: its output row
  __shared__ __attribute__((aligned(16))) float s_coef[2][FIN ? Kd : 4];      // (scale | shift) of the input transform
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const int region = blockIdx.x % PL_REGIONS, lw = (blockIdx.x / PL_REGIONS) * 4 + wid;
  const unsigned long long t_start = (DBG && pv.trace) ? __builtin_amdgcn_s_memtime() : 0ull;    // per-wave stamps: debug / trace instances only
  unsigned long long t_pro = 0ull, t_loop = 0ull, t_mark = t_start;      // cycles in pass prologues / main loops, summed over the passes
  unsigned trace_work = 0;
  int32_t(*s_idx)[64] = s_idx_all[wid];
  float in_lo = 0.f;
  if constexpr (FIN) {
    for (int c = threadIdx.x; c < 2 * Kd; c += 256) s_coef[c / Kd][c % Kd] = pv.in_coef[c];
    in_lo = pv.in_relu ? 0.f : -__builtin_inff();
  }
  const int nt_total = pv.nc_total / 16, col_tile0 = blockIdx.y * NT;
  const i32x4 srd_x = make_srd(a.X, x_bytes), srd_w = make_srd(wfrag, w_bytes);
  const int32_t* my_tiles = pv.tile_of + ((int64_t)region * PL_REGION_WAVES + lw) * (pv.d.n_pass * RS_G);
  static_assert(!DYN || (RS_G == 1 && DBG == 0), "the dynamic list hands out single tiles to production instances");
  __shared__ int s_next;                                                               // DYN: next position of the workgroup's list
  __shared__ float s_bnpos[DYN ? 2 : 1][DYN ? RS3_DYN_COLS : 1];                       // DYN: column sums / sums of squares of the tile at every list position
  const int dyn_slots = pv.d.n_pass * pv.d.G, dyn_n = 4 * dyn_slots;                   // the plan's slots per wave (laid out for pv.d.G tiles per pass)
  const int32_t* wg_tiles = pv.tile_of + ((int64_t)region * PL_REGION_WAVES + (blockIdx.x / PL_REGIONS) * 4) * dyn_slots;
  if constexpr (DYN) {
    if (threadIdx.x == 0) s_next = 0;
    for (int e = threadIdx.x; e < 2 * RS3_DYN_COLS; e += 256) s_bnpos[e / RS3_DYN_COLS][e % RS3_DYN_COLS] = 0.f;
  }
  if constexpr (DYN || FIN) __syncthreads();               // the only workgroup barrier in front of the epilogue
  int dyn_pos = 0;

  // BatchNorm statistics: lane c < 16 NT sums column c of every tile the wave stores, read back from the staging tile row by row (fixed order).
  // Two registers carried across the pass loop; the first version summed straight from the accumulators (column 16 t + li in lane (li, kk):
  // 2 NT registers), which hipcc spilled around the main loop at four waves per SIMD.
  float bn0 = 0.f, bn1 = 0.f;
#pragma nounroll
  for (int pass = 0; DYN || pass < pv.d.n_pass; ++pass) {
  int32_t dyn_tile = -1;
  if constexpr (DYN) {
    // position i of the list = slot i / 4 of wave i % 4: the deal fills slot levels heaviest first
    int i = 0;
    if (lane == 0) i = atomicAdd(&s_next, 1);
    i = __builtin_amdgcn_readfirstlane(i);
    if (i >= dyn_n) break;
    dyn_pos = i;
    dyn_tile = wg_tiles[(i & 3) * dyn_slots + (i >> 2)];
    if (dyn_tile < 0) continue;                            // a wave whose slots ended early: other waves' deeper levels may still hold tiles
  } else {
    if (my_tiles[pass * RS_G] < 0) break;                  // slots are filled front to back: an empty first slot ends the wave's list
  }
  if (pv.prio & 3) __builtin_amdgcn_s_setprio(3);
  if constexpr (DYN) bn0 = 0.f, bn1 = 0.f;                 // this tile's column sums only (kept per list position)
  // the wave's rows of the regrouped table -> LDS; per-offset tile masks in lane k of maskreg
  unsigned maskreg = 0;
  {
    const int g = lane >> 4;
    const int32_t t = DYN ? (g < 1 ? dyn_tile : -1) : (g < RS_G ? my_tiles[pass * RS_G + g] : -1);
    i32x4 e[PL_ROW / 4];
    const int64_t p = (int64_t)t * 16 + li;
    const int32_t row = t >= 0 ? pv.perm[p] : -1;
    unsigned m = row >= 0 ? (unsigned)pv.masks_p[p] : 0u;
    if (row >= 0) {
      const i32x4* rowp = reinterpret_cast<const i32x4*>(pv.tab + (int64_t)row * PL_ROW);
#pragma unroll
      for (int q = 0; q < (RS3_KMAX + 3) / 4; ++q) e[q] = rowp[q];
    } else {
#pragma unroll
      for (int q = 0; q < (RS3_KMAX + 3) / 4; ++q) e[q] = (i32x4){-1, -1, -1, -1};
    }
#pragma unroll
    for (int k = 0; k < RS3_KMAX; ++k)
      s_idx[k][lane] = e[k >> 2][k & 3];            // table order (entries >= K are -1 in the table); a reversed table is read at K-1-k in the loop
    s_idx[RS3_KMAX][lane] = row;                           // output row (-1: padding)
    if (pv.k_flip) m = __brev(m) >> (32 - a.K);
    unsigned mall = m;                                                      // FIN: offsets EVERY row of the tile has (a padding row has none: its tile always clears)
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) m |= __shfl_xor(m, off, 16);      // OR over the tile's 16 rows
    if constexpr (FIN) {
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) mall &= __shfl_xor(mall, off, 16);
    }
    unsigned mk = 0;
#pragma unroll
    for (int t2 = 0; t2 < RS_G; ++t2) {
      const unsigned tm = (unsigned)__builtin_amdgcn_readlane((int)m, 16 * t2);
      mk |= ((tm >> (lane & 31)) & 1u) << t2;
      if constexpr (FIN) {                                                  // bit 8 + t2: tile t2 has a neighbour in every row at this offset
        const unsigned ta = (unsigned)__builtin_amdgcn_readlane((int)mall, 16 * t2);
        mk |= ((ta >> (lane & 31)) & 1u) << (8 + t2);
      }
    }
    maskreg = lane < a.K ? mk : 0u;
  }
  const unsigned long long active = __ballot((maskreg & 0xffu) != 0);
  if constexpr (DBG != 0) {
    if (pv.trace) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_pro += t - t_mark, t_mark = t;
    }
  }

  f32x4 acc[RS_G][NT];
#pragma unroll
  for (int g = 0; g < RS_G; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (pv.prio & 3) __builtin_amdgcn_s_setprio(0);
  if (pv.prio & 4) {
    // Longest wave first: the wave's priority for its main loop follows its own (tile, offset) step count.  A SIMD's waves share one matrix pipe;
    // a wave left alone on it cannot fill it (it waits for its own gathers), so the launch ends sooner when the long waves are served first and the
    // short ones fill the gaps than when all run at the same rate and the long one finishes alone.
    int steps = 0;
#pragma unroll
    for (int t2 = 0; t2 < RS_G; ++t2) steps += __popcll(__ballot((maskreg >> t2) & 1u));
    const int unit = (pv.prio >> 4) > 0 ? (pv.prio >> 4) : 7 * RS_G;          // steps per priority level
    const int lvl = steps / unit;
    if (lvl >= 3) __builtin_amdgcn_s_setprio(3);
    else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
    else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  }

  if (active) {
    // Operand rings.  Rows (A): RA stages = RA - 1 steps of gathers in flight (a gathered row comes from HBM / a remote L2 line: ~2 us under load).
    // Weights (B): RB = 2 stages, ONE step ahead -- the 442 KB of fragments are shared by every wave of the launch and sit in L2, a step of four
    // waves' MFMAs (~4 k cycles) covers that latency; the third B stage (16 VGPRs at NT = 4) is what kept the two-tile instances at the
    // 128-register line with spills in every pass prologue, and is what the input transform's (FIN) coefficients now live in.
    // (History: one ring of 3 stages for both operands until round 5.  Tried on it: 5 stages for the one-tile-per-wave instance -- slower,
    // 64->64 at 66 k rows 78 -> 85 us, 64->128 28 -> 39 us: the extra dummy loads of the tail and the longer prologue cost more than the
    // lookahead buys (round 1 found the same on the narrow kernels); 4 stages for it in round 3 (128 VGPRs, no spill): 65.5 -> 67-68 us.)
    // DEEP (round 6): the one-tile instances of the 64-column kernels keep FOUR stages of both operands (three steps of loads in flight): such a
    // launch ends with its densest tiles, each a wave alone on its SIMD walking 27 offsets x KQ steps, and a step of a lone wave lasts as long as
    // its youngest operand's latency -- the weights', requested ONE step ahead (tools/conv_trace.py: ~1100-1260 cycles per step against 512 of
    // MFMAs; four stages x (1 + 4) quads = 80 VGPRs where two tiles would need 96 + 32 accumulators).  Four stages and KQ % 4 == 0 also make the
    // stage of a step its q: one offset per trip.
    constexpr bool DEEP = (SEEVCN_RS3_DEEP != 0) && (SEEVCN_RS3_OFFSET_LOOP != 0) && RS_G == 1 && NT == 4 && KQ % 4 == 0 && RS3_RA == 3 && RS3_RB == 2;
    constexpr int RA = DEEP ? 4 : RS3_RA, RB = DEEP ? 4 : RS3_RB, UNR = (RA % RB == 0) ? RA : RA * RB;
    static_assert(RA >= RB && RB >= 2, "the counted wait below is written for a weight ring no deeper than the row ring");
    f32x4 A[RA][RS_G], B[RB][NT];
    // defined here so that their live ranges start inside the pass (the asm waits below read-modify them: left undefined, hipcc keeps all
    // stage registers alive across the whole pass loop, prologue and epilogue included, and spills 55 VGPRs at four waves per SIMD)
#pragma unroll
    for (int st = 0; st < RA; ++st)
#pragma unroll
      for (int g = 0; g < RS_G; ++g) A[st][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < RB; ++st)
#pragma unroll
      for (int t = 0; t < NT; ++t) B[st][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr uint32_t OOB = 0xfffffff0u;
    // the weight loads add immediates of up to 3 KiB to their vector offset: an out-of-range value that cannot wrap around 2^32 with them
    // (the fragment buffer is a few MB)
    constexpr uint32_t WOOB = 0x80000000u;
    constexpr uint32_t WSTEP = 1024u;                                // one (offset, q, column tile) fragment
    const uint32_t wq = (uint32_t)nt_total * WSTEP;                   // q -> q + 1
    const uint32_t wlane = (uint32_t)lane * 16u;
    auto math = [&](f32x4 (&As)[RS_G], f32x4 (&Bs)[NT], const int qc, const unsigned mc, const int32_t (&fin_absent)[RS_G]) {
      // Issue order inside a step: weights of step s + 1, then rows of step s + RA - 1.  The younger of this step's operands is its weight stage
      // (issued one step ago, in front of that step's row loads): behind it in the queue are one step's row loads and one whole step,
      // NWAIT = RS_G + (RS_G + NT) loads that may stay outstanding; loads return in order, so the row stage (older) has arrived as well.
      // (RA == RB, the measurement build SEEVCN_RS3_RB=3: both of a step's stages were issued RA - 1 steps ago, whole steps only behind them.)
      // FIN: the step's coefficients (channels 16 qc + 4 kk + {0..3} of the lane's four values) and the rows' validity are LDS reads that do not
      // depend on the operands: requested here, in front of the wait, their latency hides behind it (behind it they sat on every step's critical
      // path: the two-tile forward at 139 k rows ran 127 us against 115 without the transform)
      f32x4 fin_sc, fin_sh;
      if constexpr (FIN) fin_sc = *reinterpret_cast<const f32x4*>(&s_coef[0][qc * 16 + kk * 4]), fin_sh = *reinterpret_cast<const f32x4*>(&s_coef[1][qc * 16 + kk * 4]);
      constexpr int NWAIT = RA > RB ? RS_G + (RB - 1) * (RS_G + NT) : (RA - 1) * (RS_G + NT);
#define RS3_WAIT(...) asm volatile("s_waitcnt vmcnt(%[nw])" : __VA_ARGS__ : [nw] "n"(NWAIT))
#define V(x) "+v"(x)
      if constexpr (RS_G == 4 && NT == 4) { RS3_WAIT(V(As[0]), V(As[1]), V(As[2]), V(As[3]), V(Bs[0]), V(Bs[1]), V(Bs[2]), V(Bs[3])); }
      else if constexpr (RS_G == 4 && NT == 2) { RS3_WAIT(V(As[0]), V(As[1]), V(As[2]), V(As[3]), V(Bs[0]), V(Bs[1])); }
      else if constexpr (RS_G == 4 && NT == 1) { RS3_WAIT(V(As[0]), V(As[1]), V(As[2]), V(As[3]), V(Bs[0])); }
      else if constexpr (RS_G == 3 && NT == 4) { RS3_WAIT(V(As[0]), V(As[1]), V(As[2]), V(Bs[0]), V(Bs[1]), V(Bs[2]), V(Bs[3])); }
      else if constexpr (RS_G == 3 && NT == 2) { RS3_WAIT(V(As[0]), V(As[1]), V(As[2]), V(Bs[0]), V(Bs[1])); }
      else if constexpr (RS_G == 3 && NT == 1) { RS3_WAIT(V(As[0]), V(As[1]), V(As[2]), V(Bs[0])); }
      else if constexpr (RS_G == 2 && NT == 4) { RS3_WAIT(V(As[0]), V(As[1]), V(Bs[0]), V(Bs[1]), V(Bs[2]), V(Bs[3])); }
      else if constexpr (RS_G == 2 && NT == 2) { RS3_WAIT(V(As[0]), V(As[1]), V(Bs[0]), V(Bs[1])); }
      else if constexpr (RS_G == 1 && NT == 4) { RS3_WAIT(V(As[0]), V(Bs[0]), V(Bs[1]), V(Bs[2]), V(Bs[3])); }
      else {
        static_assert(RS_G == 2 && NT == 1, "no counted wait for this (tiles per wave, column tiles) pair");
        RS3_WAIT(V(As[0]), V(As[1]), V(Bs[0]));
      }
#undef V
#undef RS3_WAIT
      // per tile: 4 passes over the NT column tiles, so that consecutive MFMAs never share an accumulator (a dependent
      // v_mfma_f32_16x16x4_f32 issues 47 cycles after its producer, an independent one after 32).
      // FIN: the transform of a tile's k-th operand register (fused multiply-add, max, and -- unless every row of the tile has this neighbour -- the
      // clearing of the rows without one) is written BEHIND the first MFMA of the register before it: the wave issues it while the matrix pipe works
      // on that MFMA instead of in front of the step's whole MFMA block (12-15 % on every forward launch when it sat there).
      auto fin1 = [&](int g, int i) {
        float v = fmaxf(__fmaf_rn(As[g][i], fin_sc[i], fin_sh[i]), in_lo);
        if (!((mc >> (8 + g)) & 1u)) v = __uint_as_float(__float_as_uint(v) & ~(uint32_t)fin_absent[g]);       // wave-uniform test
        As[g][i] = v;
      };
#pragma unroll
      for (int g = 0; g < RS_G; ++g)
        if (((mc >> g) & 1u) && !(DBG == 1 && (pv.debug & 4))) {
          if constexpr (FIN) fin1(g, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(As[g][i], Bs[0][i], acc[g][0], 0, 0, 0);
            if constexpr (FIN) {
              if (i + 1 < 4) fin1(g, i + 1);
            }
#pragma unroll
            for (int t = 1; t < NT; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(As[g][i], Bs[t][i], acc[g][t], 0, 0, 0);
          }
        }
    };
    if constexpr ((SEEVCN_RS3_OFFSET_LOOP != 0) && KQ >= 2 && ((RA == 3 && RB == 2) || DEEP)) {
      // The loop by OFFSET (round 6).  The first form below keeps three iterators (row loads, weight loads, compute) that each test "was this
      // the offset's last 16-channel step" in EVERY step: ~30 scalar instructions and 6-8 taken branches between two MFMA blocks (hipcc keeps
      // a step's MFMAs together, the bookkeeping is not interleaved with them).  With four waves on a SIMD the others' MFMAs cover that; a wave
      // left ALONE does not: per-wave stamps (tools/conv_trace.py, profiles/r06_conv_trace_raw.txt) show every 64 -> 64 launch of <= 74 k rows
      // ending with its 27-offset tiles, each walking 108 steps at ~1080 cycles where its 16 MFMAs need 512 -- the launch lasts as long as
      // that wave (144 k cycles where the busiest SIMD needs 113 k, the mean one 89 k).  Here an iteration is one offset: its KQ steps are
      // unrolled with compile-time q, the rows of step s + 2 and the weights of step s + 1 are "this offset's q + 2 / q + 1 or the next
      // offset's q + 2 - KQ / 0" decided at compile time, and the list advances ONCE per offset (next offset, its gathered-row offsets from
      // LDS, its weight base, its tile mask).  Three offsets per trip make the ring stages compile-time too (KQ steps per offset, 3 row stages).
      // Same loads in the same order, same counted wait, same MFMA order: bit-identical results.
      unsigned long long rest = active;                              // offsets behind k_n
      int k_c = __ffsll((long long)rest) - 1;
      rest &= rest - 1;
      int k_n = rest ? __ffsll((long long)rest) - 1 : -1;
      rest &= rest - 1;
      auto rows_of = [&](int k, uint32_t (&ro)[RS_G]) {              // k >= 0: byte offset of the lane's gathered row (+ its 4-channel column), OOB without a neighbour
#pragma unroll
        for (int g = 0; g < RS_G; ++g) {
          const int32_t j = s_idx[pv.k_flip ? a.K - 1 - k : k][g * 16 + li];
          ro[g] = j >= 0 ? (uint32_t)(j * (Kd * 4) + kk * 16) : OOB;
          if constexpr (DBG == 1) {
            if (pv.debug & 1) ro[g] = OOB;
            else if ((pv.debug & 16) && j >= 0) ro[g] = 0u;
          }
        }
      };
      auto wbase = [&](int k) { return (uint32_t)(k * KQ * nt_total + col_tile0) * WSTEP; };
      uint32_t ro_c[RS_G], ro_n[RS_G], ro_t[RS_G];
      rows_of(k_c, ro_c);
#pragma unroll
      for (int g = 0; g < RS_G; ++g) ro_n[g] = OOB;
      if (k_n >= 0) rows_of(k_n, ro_n);
      uint32_t wv_c = wlane;
      if constexpr (DBG == 1) {
        if (pv.debug & 2) wv_c = WOOB;
        else if (pv.debug & 8) wv_c = 0u;
      }
      uint32_t wv_n = k_n >= 0 ? wv_c : WOOB;                         // past the list's end: out-of-range dummies (exactly NT + RS_G loads per step, always)
      uint32_t sw_c = wbase(k_c), sw_n = k_n >= 0 ? wbase(k_n) : 0u;
      unsigned mc = (unsigned)__builtin_amdgcn_readlane((int)maskreg, k_c);
      auto load_a = [&](f32x4 (&As)[RS_G], const uint32_t (&ro)[RS_G], uint32_t sa) {
#pragma unroll
        for (int g = 0; g < RS_G; ++g) As[g] = buf_load_b128_s<0>(srd_x, ro[g], sa);
      };
      auto load_b = [&](f32x4 (&Bs)[NT], uint32_t wv, uint32_t sw) {
        if constexpr (NT >= 1) Bs[0] = buf_load_b128_w<0>(srd_w, wv, sw);
        if constexpr (NT >= 2) Bs[1] = buf_load_b128_w<1024>(srd_w, wv, sw);
        if constexpr (NT >= 3) Bs[2] = buf_load_b128_w<2048>(srd_w, wv, sw);
        if constexpr (NT >= 4) Bs[3] = buf_load_b128_w<3072>(srd_w, wv, sw);
      };
      constexpr int LA = RA - 1, LB = RB - 1;                         // steps of row / weight loads in flight
      static_assert(LA <= KQ && LB <= LA, "a step's loads reach at most into the next offset");
      constexpr int TRIP_STEPS = (KQ % RA == 0 && KQ % RB == 0) ? KQ : ((KQ * RA) % RB == 0 ? KQ * RA : KQ * RA * RB);
      constexpr int NTRIP = TRIP_STEPS / KQ;                          // offsets per trip: the ring stages of a step are compile-time
      // ring fill, in the loop's own issue order (virtual steps -LA .. -1: weights of step v + LB, then rows of step v + LA; all in the first offset)
#pragma unroll
      for (int v = -LA; v < 0; ++v) {
        if (v + LB >= 0) load_b(B[(v + LB) % RB], wv_c, sw_c + (uint32_t)(v + LB) * wq);
        load_a(A[(v + LA) % RA], ro_c, 64u * (uint32_t)(v + LA));
      }
      for (bool more = true; more;) {
#pragma unroll
        for (int u = 0; u < NTRIP; ++u) {
          int32_t absent[RS_G];
#pragma unroll
          for (int g = 0; g < RS_G; ++g) absent[g] = FIN ? (ro_c[g] == OOB ? -1 : 0) : 0;     // all-ones: the row has no neighbour at this offset
          int k_t = -1;
#pragma unroll
          for (int q = 0; q < KQ; ++q) {
            const int st = u * KQ + q;                               // step inside the trip (compile-time after unrolling)
            if (q + LB < KQ) load_b(B[(st + LB) % RB], wv_c, sw_c + (uint32_t)(q + LB) * wq);
            else load_b(B[(st + LB) % RB], wv_n, sw_n + (uint32_t)(q + LB - KQ) * wq);
            if (q + LA < KQ) load_a(A[(st + LA) % RA], ro_c, 64u * (uint32_t)(q + LA));
            else load_a(A[(st + LA) % RA], ro_n, 64u * (uint32_t)(q + LA - KQ));
            if (q == KQ - 1) {
              // the offset behind the next one: its rows' offsets are requested from LDS here, in front of this step's MFMAs, and first used
              // KQ - LA steps into the next offset
              k_t = rest ? __ffsll((long long)rest) - 1 : -1;
              rest &= rest - 1;
#pragma unroll
              for (int g = 0; g < RS_G; ++g) ro_t[g] = OOB;
              if (k_t >= 0) rows_of(k_t, ro_t);
            }
            math(A[st % RA], B[st % RB], q, mc, absent);
          }
          k_c = k_n, k_n = k_t;
          if (k_c < 0) {
            more = false;
            break;
          }
#pragma unroll
          for (int g = 0; g < RS_G; ++g) ro_c[g] = ro_n[g], ro_n[g] = ro_t[g];
          sw_c = sw_n, sw_n = k_n >= 0 ? wbase(k_n) : 0u;
          wv_n = k_n >= 0 ? wv_c : WOOB;
          mc = (unsigned)__builtin_amdgcn_readlane((int)maskreg, k_c);
        }
      }
    } else {
    // Row-load iterator (RA - 1 steps ahead of the compute iterator).  Per offset: rowoff[g] = byte offset of the lane's gathered row (+ its
    // 4-channel column), or an out-of-range value when the row has no neighbour there / the list has ended.  Per 16-channel step the loads then
    // need NO vector arithmetic: rows at rowoff + (scalar) 64 * q, weights at one constant per-lane offset + (scalar) fragment base of
    // (offset, q) + (immediate) 1 KiB * column tile.  (The first version recomputed every load's offset each step: 74 scalar + 25 vector
    // instructions per step next to its 32 MFMAs.)
    unsigned long long la = active;
    int kl = __ffsll((long long)la) - 1, ql = 0;
    uint32_t rowoff[RS_G];
    uint32_t wvoff = wlane;                                          // OOB once the list has ended (dummy loads)
    auto read_j = [&]() {
#pragma unroll
      for (int g = 0; g < RS_G; ++g) {
        const int32_t j = s_idx[pv.k_flip ? a.K - 1 - kl : kl][g * 16 + li];
        rowoff[g] = j >= 0 ? (uint32_t)(j * (Kd * 4) + kk * 16) : OOB;
        if constexpr (DBG == 1) {
          if (pv.debug & 1) rowoff[g] = OOB;
          else if ((pv.debug & 16) && j >= 0) rowoff[g] = 0u;
        }
      }
    };
    read_j();
    if constexpr (DBG == 1) {
      if (pv.debug & 2) wvoff = WOOB;
      else if (pv.debug & 8) wvoff = 0u;
    }
    // running scalar offsets: sa = 64 * q (rows), sw = fragment base of (offset, q) (weights)
    uint32_t sa = 0u;
    auto issue_a = [&](f32x4 (&As)[RS_G]) {                          // exactly RS_G loads, always
#pragma unroll
      for (int g = 0; g < RS_G; ++g) As[g] = buf_load_b128_s<0>(srd_x, rowoff[g], sa);
      if (kl >= 0) {
        sa += 64u;
        if (++ql == KQ) {
          ql = 0, sa = 0u;
          la &= la - 1;
          kl = la ? __ffsll((long long)la) - 1 : -1;
          if (kl >= 0) read_j();
          else {                                                   // the list has ended: every further load is an out-of-range dummy
#pragma unroll
            for (int g = 0; g < RS_G; ++g) rowoff[g] = OOB;
          }
        }
      }
    };
    // weight-load iterator (one step ahead)
    unsigned long long lb = active;
    int kb = __ffsll((long long)lb) - 1, qb = 0;
    uint32_t sw = (uint32_t)((kb < 0 ? 0 : kb) * KQ * nt_total + col_tile0) * WSTEP;
    auto issue_b = [&](f32x4 (&Bs)[NT]) {                            // exactly NT loads, always
      if constexpr (NT >= 1) Bs[0] = buf_load_b128_w<0>(srd_w, wvoff, sw);
      if constexpr (NT >= 2) Bs[1] = buf_load_b128_w<1024>(srd_w, wvoff, sw);
      if constexpr (NT >= 3) Bs[2] = buf_load_b128_w<2048>(srd_w, wvoff, sw);
      if constexpr (NT >= 4) Bs[3] = buf_load_b128_w<3072>(srd_w, wvoff, sw);
      if (kb >= 0) {
        sw += wq;
        if (++qb == KQ) {
          qb = 0;
          lb &= lb - 1;
          kb = lb ? __ffsll((long long)lb) - 1 : -1;
          if (kb >= 0) sw = (uint32_t)(kb * KQ * nt_total + col_tile0) * WSTEP;
          else wvoff = WOOB;
        }
      }
    };
    // compute iterator
    unsigned long long ca = active;
    int kc = __ffsll((long long)ca) - 1, qc = 0;
    unsigned mc = (unsigned)__builtin_amdgcn_readlane((int)maskreg, kc);
    auto compute = [&](f32x4 (&As)[RS_G], f32x4 (&Bs)[NT]) {
      int32_t fin_absent[RS_G];
#pragma unroll
      for (int g = 0; g < RS_G; ++g) fin_absent[g] = 0;
      if constexpr (FIN) {
#pragma unroll
        for (int g = 0; g < RS_G; ++g) fin_absent[g] = s_idx[pv.k_flip ? a.K - 1 - kc : kc][g * 16 + li] >> 31;     // all-ones: no neighbour
      }
      math(As, Bs, qc, mc, fin_absent);
      if (++qc == KQ) {
        qc = 0;
        ca &= ca - 1;
        kc = ca ? __ffsll((long long)ca) - 1 : -1;
        if (kc >= 0) mc = (unsigned)__builtin_amdgcn_readlane((int)maskreg, kc);
      }
    };
    // the steps in front of the first compute, in the loop's own issue order (virtual steps -(RA - 1) .. -1)
#pragma unroll
    for (int s0 = -(RA - 1); s0 < 0; ++s0) {
      if (s0 + RB - 1 >= 0) issue_b(B[(s0 + RB - 1) % RB]);
      issue_a(A[(s0 + RA - 1) % RA]);
    }
    for (bool more = true; more;) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        issue_b(B[(u + RB - 1) % RB]);
        issue_a(A[(u + RA - 1) % RA]);
        compute(A[u % RA], B[u % RB]);
        if (kc < 0) {
          more = false;
          break;
        }
      }
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // retire the dummy loads before the registers are reused
  }
  if ((pv.prio & 3) == 2) __builtin_amdgcn_s_setprio(3);
  if (DBG && pv.trace) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    t_loop += t - t_mark, t_mark = t;
    for (int k = 0; k < a.K; ++k) trace_work += __popc((unsigned)__builtin_amdgcn_readlane((int)maskreg, k));
  }
  // D layout (16x16): col = lane&15, row = 4*(lane>>4) + reg
  const bool plain_out = !a.bias && !a.scale && !a.residual && !a.relu;       // the training layers: BatchNorm follows, nothing fused here
  // An opaque copy of the lane id for the epilogue: its addresses (staging tile, output columns) are invariant over the pass loop, hipcc
  // hoists them in front of it and then SPILLS them across the main loop (21 VGPRs, 43 MB of scratch traffic per launch by PMC) -- derived
  // from a value defined here they are computed where they are used.
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int li_e = lane_e & 15, kk_e = lane_e >> 4;
  if (plain_out) {
    // Whole rows out: a tile goes through wave-private LDS (rows 0..16 of the neighbour-index block, free now; row RS3_KMAX = the output rows
    // stays) so that the 16 x NT lanes of a row store its 64 * NT bytes with one instruction.  Stored straight from the accumulators a row left
    // as four 64-byte pieces in four instructions: 58 MB written for a 35.7 MB output on the 64-channel layers (PMC WRITE_SIZE, round 2).
    float* T = reinterpret_cast<float*>(&s_idx[0][0]);
    constexpr int TP = NT * 16 + 4;                                 // pitch: 68 floats at NT = 4 (17 index rows of 64 ints hold 16 of them)
    static_assert(16 * TP <= 17 * 64, "the staging tile must fit under the output-row line of the index block");
#pragma unroll
    for (int g = 0; g < RS_G; ++g) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          T[(kk_e * 4 + r) * TP + t * 16 + li_e] = acc[g][t][r];
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (pv.bn_partial && (NT == 4 || lane_e < NT * 16)) {
        if (pv.bn_x) {
          // backward sums of the BatchNorm whose output gradient this tile is: x rows of the tile (whole 64 NT-byte runs, one per row)
          const int col = col_tile0 * 16 + lane_e;
          const float mean = pv.bn_mean[col], istd = pv.bn_istd[col];
          const float sc = istd * (pv.bn_gamma ? pv.bn_gamma[col] : 1.f), sh = bn_shift(mean, sc, pv.bn_beta ? pv.bn_beta[col] : 0.f);
          float xv[16];
#pragma unroll
          for (int rw = 0; rw < 16; ++rw) {
            const int row = __builtin_amdgcn_readfirstlane(s_idx[RS3_KMAX][g * 16 + rw]);
            xv[rw] = row >= 0 ? pv.bn_x[(int64_t)row * pv.nc_total + col] : 0.f;
          }
#pragma unroll
          for (int rw = 0; rw < 16; ++rw) {
            float d = T[rw * TP + lane_e];                                // padding rows hold zeros
            if (pv.bn_relu) d = bn_act(xv[rw], sc, sh) > 0.f ? d : 0.f;
            bn0 += d, bn1 += d * ((xv[rw] - mean) * istd);
          }
        } else {
#pragma unroll
          for (int rw = 0; rw < 16; ++rw) {
            const float v = T[rw * TP + lane_e];                          // padding rows hold zeros
            bn0 += v, bn1 += v * v;
          }
        }
        if constexpr (DYN) s_bnpos[0][dyn_pos * (NT * 16) + lane_e] = bn0, s_bnpos[1][dyn_pos * (NT * 16) + lane_e] = bn1;
      }
      constexpr int C4N = NT * 4;                                     // 16-byte pieces per row
#pragma unroll
      for (int i = 0; i < (16 * C4N + 63) / 64; ++i) {
        const int f = lane_e + 64 * i, rw = f / C4N, c4 = f % C4N;
        if (16 * C4N % 64 == 0 || f < 16 * C4N) {
          const int64_t row = s_idx[RS3_KMAX][g * 16 + rw];
          if (row >= 0) *reinterpret_cast<f32x4*>(a.Y + row * pv.nc_total + col_tile0 * 16 + c4 * 4) = *reinterpret_cast<const f32x4*>(T + rw * TP + c4 * 4);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // the next pass refills the index block
  } else {
#pragma unroll
    for (int g = 0; g < RS_G; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = s_idx[RS3_KMAX][g * 16 + kk_e * 4 + r];
        if (row < 0) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int col = (col_tile0 + t) * 16 + li_e;
          a.Y[row * pv.nc_total + col] = conv_epilogue(acc[g][t][r], col, row, a);
        }
      }
  }
    if (DBG && pv.trace) t_mark = __builtin_amdgcn_s_memtime();          // the pass's epilogue ends here: (life - prologues - loops) is epilogue time
  }   // pass
  if (pv.bn_partial) {
    // BatchNorm statistics of this launch's output: lane c holds column c of its wave -> the workgroup (4 waves, fixed order) -> one partial
    // per workgroup and column, combined by k_bn_finalize in a fixed order
    if constexpr (DYN) {
      __syncthreads();                                     // every wave has left the list: all positions are final (untouched ones hold zeros)
      for (int e = threadIdx.x; e < 2 * NT * 16; e += 256) {
        const int which = e / (NT * 16), c = e % (NT * 16);
        float v = 0.f;
        for (int pos = 0; pos < dyn_n; ++pos) v += s_bnpos[which][pos * (NT * 16) + c];      // list order: the same sum whichever wave took which tile
        pv.bn_partial[((size_t)blockIdx.x * 2 + which) * pv.nc_total + col_tile0 * 16 + c] = v;
      }
    } else {
    __shared__ float s_bn[4][2][64];
    if (NT == 4 || lane < NT * 16) s_bn[wid][0][lane] = bn0, s_bn[wid][1][lane] = bn1;
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * NT * 16; e += 256) {
      const int which = e / (NT * 16), c = e % (NT * 16);
      const float v = (s_bn[0][which][c] + s_bn[1][which][c]) + (s_bn[2][which][c] + s_bn[3][which][c]);
      pv.bn_partial[((size_t)blockIdx.x * 2 + which) * pv.nc_total + col_tile0 * 16 + c] = v;
    }
    }
  }
  if (DBG && pv.trace && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned work = trace_work;
    unsigned long long* o = pv.trace + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wid) * 8;
    o[0] = t_start, o[1] = t_pro, o[2] = t_loop, o[3] = t_end, o[4] = hw, o[5] = xcc, o[6] = work, o[7] = ((unsigned long long)blockIdx.x << 8) | wid;
  }
}

// shapes the plan kernel is instantiated for: C_in in {16, 32, 64, 128}, C_out in {16, 32, 64} or a multiple of 64, K <= 27
static bool rs3_applies(int K, int Kd, int Nc) {
  if (K > RS3_KMAX || K < 1) return false;
  const bool kd_ok = Kd == 16 || Kd == 32 || Kd == 64 || Kd == 128;
  const bool nc_ok = Nc == 16 || Nc == 32 || (Nc >= 64 && Nc % 64 == 0 && Nc <= 512);
  return kd_ok && nc_ok;
}
// workgroups along x of every planned launch = BatchNorm partials its epilogue writes
extern "C" int sv_conv_planned_partials(void) { return PL_REGIONS * PL_REGION_WAVES / 4; }

extern "C" int sv_conv_mfma_kernel_applies(int K, int Kd, int Nc, int64_t n_src) {
  // the gathers address X through a 32-bit buffer descriptor
  return (rs3_applies(K, Kd, Nc) && (uint64_t)(n_src < 0 ? 0 : n_src) * Kd * 4 < 0xfffffff0ull) ? 1 : 0;
}

template <int NT, int KQ>
static void launch_rs3_g(const ConvArgs& a, const PlanView& pv, const float* wfrag, uint32_t xb, uint32_t wb, dim3 grid, hipStream_t st) {
  // tiles per pass: conv_tiles_per_wave (1 or 2 for the 64-column kernels, 4 for the narrow ones); the measurement switches of
  // SEEVCN_RS3_DEBUG live in instances of their own so that the production loop carries none of their tests
  if (pv.debug) {
    if constexpr (NT == 4) {
      if (pv.d.G == 1) hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 1, 1>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
      else hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 2, 1>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
    } else {
      hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 4, 1>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
    }
    return;
  }
  if (pv.trace) {                      // the production loop with time stamps
    if constexpr (NT == 4) {
      if (pv.d.G == 1) hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 1, 2>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
      else hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 2, 2>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
    } else {
      hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 4, 2>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
    }
    return;
  }
  if constexpr (NT == 4) {
    // SEEVCN_RS3_DYN=1 (A/B, off by default): the workgroup-local dynamic list (one tile per pass) for the 64-column kernels whenever the workgroup's
    // list fits the per-position sums.  Measured (round 5, same box, profiles/r05_dyn_ab.txt): 64 -> 64 at 139 k rows 114.9 -> 119.9 us, the other layers
    // +-1 %, the step 4.05 ms either way -- equal wave lengths buy nothing here: a one-tile pass has half the MFMA work in flight per gather latency
    // (1024 pipe cycles against 2048 with two tiles), and the 66 k-row layers have one tile per wave whoever takes it.
    static const int dyn_env = getenv("SEEVCN_RS3_DYN") ? atoi(getenv("SEEVCN_RS3_DYN")) : 0;
    if (dyn_env && 4 * pv.d.n_pass * pv.d.G * NT * 16 <= RS3_DYN_COLS) {
      if (pv.in_coef) hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 1, 0, true, true>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
      else hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 1, 0, false, true>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
      return;
    }
  }
  if (pv.in_coef) {                    // BatchNorm (+ ReLU) of the layer below applied on load
    if constexpr (NT == 4) {
      if (pv.d.G == 1) hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 1, 0, true>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
      else hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 2, 0, true>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
    } else {
      hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 4, 0, true>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
    }
    return;
  }
  if constexpr (NT == 4) {
    if (pv.d.G == 1) hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 1>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
    else hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 2>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
  } else {
    hipLaunchKernelGGL((k_spconv_rs3<NT, KQ, 4>), grid, dim3(256), 0, st, a, pv, wfrag, xb, wb);
  }
}
template <int NT>
static void launch_rs3_kq(const ConvArgs& a, const PlanView& pv, const float* wfrag, uint32_t xb, uint32_t wb, dim3 grid, hipStream_t st) {
  switch (a.Kd / 16) {
    case 1: launch_rs3_g<NT, 1>(a, pv, wfrag, xb, wb, grid, st); break;
    case 2: launch_rs3_g<NT, 2>(a, pv, wfrag, xb, wb, grid, st); break;
    case 4: launch_rs3_g<NT, 4>(a, pv, wfrag, xb, wb, grid, st); break;
    default: launch_rs3_g<NT, 8>(a, pv, wfrag, xb, wb, grid, st); break;
  }
}

// measurement only: per-wave time stamps of the next planned launches go to `buf` (8 x uint64 per wave slot: grid.x * grid.y * 4 slots); null = off
static unsigned long long* g_conv_trace = nullptr;
extern "C" int sv_debug_conv_trace(void* buf) {
  g_conv_trace = static_cast<unsigned long long*>(buf);
  return SV_OK;
}

struct BnBwdView {            // the BatchNorm whose output gradient a data-gradient launch produces (PlanView's bn_* fields)
  const float *x, *mean, *istd, *gamma, *beta;
  int relu;
};
static int conv_planned(const float* X, int64_t n_src, const int32_t* table_rows, const int32_t* perm, const int32_t* masks_p, const int32_t* tile_of,
                        int tiles_per_wave, const float* wfrag, float* Y, int64_t n_rows, int K, int Kd, int Nc, const float* bias, const float* scale,
                        const float* shift, const float* residual, int relu, int table_k_reversed, float* bn_partial, const BnBwdView* bnb, void* stream) {
  const InNorm in = take_input_norm();
  SV_CHECK_ARG(n_rows >= 0 && K > 0 && Kd > 0 && Nc > 0, "sparse_conv (planned): bad sizes");
  SV_CHECK_ARG(!bn_partial || (!bias && !scale && !residual && !relu), "sparse_conv (planned): BatchNorm partial sums are made by the plain epilogue only");
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(X && table_rows && perm && masks_p && tile_of && wfrag && Y, "sparse_conv (planned): null pointer");
  SV_CHECK_ARG((scale == nullptr) == (shift == nullptr), "sparse_conv: scale and shift go together");
  SV_CHECK_ARG(sv_conv_mfma_kernel_applies(K, Kd, Nc, n_src), "sparse_conv (planned): no MFMA kernel for K %d, C_in %d, C_out %d, %lld source rows "
               "(ask sv_conv_mfma_kernel_applies first)", K, Kd, Nc, (long long)n_src);
  SV_CHECK_ARG(tiles_per_wave == conv_tiles_per_wave(n_rows, Kd, Nc), "sparse_conv (planned): tiles_per_wave must be sv_conv_tiles_per_wave(n_rows, Kd, Nc)");
  SV_CHECK_ARG((uintptr_t)X % 16 == 0 && (uintptr_t)wfrag % 16 == 0 && (uintptr_t)table_rows % 16 == 0, "sparse_conv (planned): 16-byte alignment");
  ConvArgs a{X, nullptr, nullptr, Y, bias, scale, shift, residual, relu, n_rows, K, Kd, Nc};
  PlanView pv;
  pv.tab = table_rows, pv.perm = perm, pv.masks_p = masks_p, pv.tile_of = tile_of, pv.d = plan_dims(n_rows, tiles_per_wave), pv.k_flip = table_k_reversed ? 1 : 0, pv.nc_total = Nc, pv.bn_partial = bn_partial;
  pv.bn_x = pv.bn_mean = pv.bn_istd = pv.bn_gamma = pv.bn_beta = nullptr, pv.bn_relu = 0;
  if (bnb) pv.bn_x = bnb->x, pv.bn_mean = bnb->mean, pv.bn_istd = bnb->istd, pv.bn_gamma = bnb->gamma, pv.bn_beta = bnb->beta, pv.bn_relu = bnb->relu;
  static const int debug = getenv("SEEVCN_RS3_DEBUG") ? atoi(getenv("SEEVCN_RS3_DEBUG")) : 0;
  pv.debug = debug;
  static const int prio = getenv("SEEVCN_RS3_PRIO") ? atoi(getenv("SEEVCN_RS3_PRIO")) : 0;
  pv.prio = prio;
  pv.trace = g_conv_trace;
  pv.in_coef = in.coef, pv.in_relu = in.relu;
  SV_CHECK_ARG(!in.coef || (!pv.debug && !pv.trace && (uintptr_t)in.coef % 16 == 0), "sparse_conv (planned): an input transform needs a production instance and 16-byte aligned coefficients");
  const int nc_blk = Nc > 64 ? 64 : Nc;
  const dim3 grid((unsigned)(PL_REGIONS * PL_REGION_WAVES / 4), (unsigned)(Nc / nc_blk));
  const uint32_t xb = (uint32_t)((uint64_t)n_src * Kd * 4), wb = (uint32_t)((uint64_t)K * Nc * Kd * 4);
  hipStream_t st = sv_stream(stream);
  switch (nc_blk / 16) {
    case 1: launch_rs3_kq<1>(a, pv, wfrag, xb, wb, grid, st); break;
    case 2: launch_rs3_kq<2>(a, pv, wfrag, xb, wb, grid, st); break;
    default: launch_rs3_kq<4>(a, pv, wfrag, xb, wb, grid, st); break;
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_sparse_conv_gather_gemm_planned(const float* X, int64_t n_src, const int32_t* table_rows, const int32_t* perm, const int32_t* masks_p,
                                                  const int32_t* tile_of, int tiles_per_wave,
                                                  const float* wfrag, float* Y, int64_t n_rows, int K, int Kd, int Nc, const float* bias,
                                                  const float* scale, const float* shift, const float* residual, int relu, int table_k_reversed,
                                                  float* bn_partial, void* stream) {
  return conv_planned(X, n_src, table_rows, perm, masks_p, tile_of, tiles_per_wave, wfrag, Y, n_rows, K, Kd, Nc, bias, scale, shift, residual, relu,
                      table_k_reversed, bn_partial, nullptr, stream);
}

// The data gradient of a layer whose INPUT came out of a BatchNorm (+ReLU): Y (n_rows, Nc) is the gradient w.r.t. that BatchNorm's output, and
// the launch's epilogue also leaves the two per-channel sums of the BatchNorm's backward -- sum(dy * branch) and sum(dy * branch * xhat), branch
// = the forward's ReLU decision recomputed from bn_x with the forward's own expression -- as sv_conv_planned_partials() per-workgroup partials in
// bn_partial, laid out like sv_batchnorm_relu_backward's scratch: sv_batchnorm_relu_backward_partial starts at the combine.
extern "C" int sv_sparse_conv_dgrad_planned_bn(const float* dZ, int64_t n_src, const int32_t* table_rows, const int32_t* perm, const int32_t* masks_p,
                                               const int32_t* tile_of, int tiles_per_wave, const float* wfrag, float* dY, int64_t n_rows, int K, int Kd,
                                               int Nc, int table_k_reversed, const float* bn_x, const float* bn_mean, const float* bn_invstd,
                                               const float* bn_gamma, const float* bn_beta, int bn_relu, float* bn_partial, void* stream) {
  SV_CHECK_ARG(bn_x && bn_mean && bn_invstd && bn_partial, "sparse_conv_dgrad_planned_bn: null pointer");
  const BnBwdView bnb{bn_x, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_relu ? 1 : 0};
  return conv_planned(dZ, n_src, table_rows, perm, masks_p, tile_of, tiles_per_wave, wfrag, dY, n_rows, K, Kd, Nc, nullptr, nullptr, nullptr, nullptr, 0,
                      table_k_reversed, bn_partial, &bnb, stream);
}

// Generic VALU path for channel counts the MFMA tiling does not cover (e.g. the C_in = 3 input layer):
// one thread per (row, 4 output columns), weights read through L1/L2.
__global__ __launch_bounds__(256) void k_spconv_valu(ConvArgs a) {
  const int nq = (a.Nc + 3) / 4;
  const int64_t total = a.n_rows * nq;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / nq;
    const int n0 = (int)(idx - row * nq) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < a.K; ++k) {
      const int32_t j = a.nbr[(int64_t)k * a.n_rows + row];
      if (j < 0) continue;
      const float* x = a.X + (int64_t)j * a.Kd;
      for (int u = 0; u < 4; ++u) {
        if (n0 + u >= a.Nc) break;
        const float* w = a.Wt + ((int64_t)k * a.Nc + n0 + u) * a.Kd;
        float s = acc[u];
        for (int c = 0; c < a.Kd; ++c) s = fmaf(x[c], w[c], s);
        acc[u] = s;
      }
    }
    for (int u = 0; u < 4 && n0 + u < a.Nc; ++u) a.Y[row * a.Nc + n0 + u] = conv_epilogue(acc[u], n0 + u, row, a);
  }
}

// Input layer (C_in = 3 or 4 point features -> 16 channels, spconv_backbone.py:77-81): HBM-bound -- 4*K bytes of neighbour table and
// 4*Nc bytes of output per row against 2*K*Kd*Nc flops.  Same thread mapping and summation order as k_spconv_valu (bit-identical
// results); the weights (K*Nc*Kd floats, 5 KB) are staged in LDS and the K table reads of a row are issued 9 at a time.
constexpr int SC_SMALL_KD = 4, SC_SMALL_LDS = 27 * 32 * SC_SMALL_KD;
template <int KD>
__global__ __launch_bounds__(256) void k_spconv_small_cin(ConvArgs a) {
  __shared__ float s_w[SC_SMALL_LDS];
  for (int i = threadIdx.x; i < a.K * a.Nc * KD; i += 256) s_w[i] = a.Wt[i];
  __syncthreads();
  const int nq = (a.Nc + 3) / 4;
  const int64_t total = a.n_rows * nq;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / nq;
    const int n0 = (int)(idx - row * nq) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < a.K; k0 += 9) {
      int32_t j[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) j[u] = k0 + u < a.K ? a.nbr[(int64_t)(k0 + u) * a.n_rows + row] : -1;
      float x[9][KD];
#pragma unroll
      for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int c = 0; c < KD; ++c) x[u][c] = j[u] >= 0 ? a.X[(int64_t)j[u] * KD + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        if (j[u] < 0) continue;
        const float* w = s_w + (size_t)((k0 + u) * a.Nc + n0) * KD;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (n0 + v >= a.Nc) break;
          float t = acc[v];
#pragma unroll
          for (int c = 0; c < KD; ++c) t = fmaf(x[u][c], w[v * KD + c], t);
          acc[v] = t;
        }
      }
    }
    for (int v = 0; v < 4 && n0 + v < a.Nc; ++v) a.Y[row * a.Nc + n0 + v] = conv_epilogue(acc[v], n0 + v, row, a);
  }
}

// Plain entry: packed (K, Nc, Kd) weights, k-major table, no plan -- the register-stationary MFMA kernel for channel multiples of 16, the
// small-C_in kernel for the 3 / 4-channel input layer, the VALU kernel for everything else.
extern "C" int sv_sparse_conv_gather_gemm(const float* X, int64_t n_src, const int32_t* nbr, const float* Wt, float* Y,
                                          int64_t n_rows, int K, int Kd, int Nc, const float* bias, const float* scale, const float* shift,
                                          const float* residual, int relu, void* stream) {
  SV_CHECK_ARG(!take_input_norm().coef, "sparse_conv: the plain entry takes no input transform (sv_conv_next_input_norm is for the planned kernel and the weight gradients)");
  SV_CHECK_ARG(n_rows >= 0 && K > 0 && Kd > 0 && Nc > 0, "sparse_conv: bad sizes");
  if (n_rows == 0) return SV_OK;
  SV_CHECK_ARG(X && nbr && Wt && Y, "sparse_conv: null pointer");
  SV_CHECK_ARG((scale == nullptr) == (shift == nullptr), "sparse_conv: scale and shift go together");
  (void)n_src;
  ConvArgs a{X, nbr, Wt, Y, bias, scale, shift, residual, relu, n_rows, K, Kd, Nc};
  hipStream_t st = sv_stream(stream);
  const int nt = Nc / 16;
  const bool mfma_ok = (Kd % 16 == 0) && (Nc % 16 == 0) && (nt == 1 || nt == 2 || nt == 4 || nt == 8) &&
                       ((uintptr_t)X % 16 == 0) && ((uintptr_t)Wt % 16 == 0);
  if (mfma_ok && try_launch_rs(a, st) == 0) {
    SV_LAUNCH_CHECK();
    return SV_OK;
  }
  if ((Kd == 3 || Kd == 4) && K * Nc * Kd <= SC_SMALL_LDS) {
    const dim3 grid(sv_grid_1d(n_rows * ((Nc + 3) / 4), 256, 256 * 8));
    if (Kd == 3) hipLaunchKernelGGL(k_spconv_small_cin<3>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_spconv_small_cin<4>, grid, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(k_spconv_valu, dim3(sv_grid_1d(n_rows * ((Nc + 3) / 4), 256, 256 * 16)), dim3(256), 0, st, a);
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: dW[k][c][n] = sum_o X[nbr[k][o]][c] * dY[o][n]   (reduction over rows)
// Stage 1: each workgroup reduces a chunk of rows for one offset k into a partial (Cin x Cout) slab
//          (MFMA 16x16x4 with the row index as the contraction dimension); stage 2 sums the slabs in a
//          fixed order -> bitwise reproducible, no atomics.
// ------------------------------------------------------------------------------------------------
constexpr int WG_CHUNK_MAX = 4096;  // rows per stage-1 workgroup (upper bound; sized per launch to fill the chip)
constexpr int WG_SUB = 4;           // 64-row blocks a wave compacts per pass
constexpr int WG_DEPTH = 4;         // operand ring depth (MFMA steps)

struct WgradArgs;
static int wgrad_chunk_rows(int64_t n_rows, int K, int groups, int Cin) {
  // aim at ~2048 workgroups (4096 for the narrow layers, whose workgroups are short: measured 16->16 47 -> 36 us, 32->32 112 -> 98 us; the
  // 64-channel layers lose with more, 156 -> 167 us): chunk = n_rows*K*groups/target rounded up to one pass of the four waves
  constexpr int64_t pass = 256 * WG_SUB;
  static const int64_t forced = getenv("SEEVCN_WGRAD_WGS") ? atoll(getenv("SEEVCN_WGRAD_WGS")) : 0;      // measurement switch
  const int64_t target = forced > 0 ? forced : (Cin <= 32 ? 4096 : 2048);
  int64_t c = (n_rows * K * groups + target - 1) / target;
  c = (c + pass - 1) / pass * pass;
  if (c < pass) c = pass;
  if (c > WG_CHUNK_MAX) c = WG_CHUNK_MAX;
  return (int)c;
}

struct WgradArgs {
  const float* X;        // (n_src, Cin)
  const int32_t* nbr;    // (K, n_rows)
  const float* dY;       // (n_rows, Cout)
  float* partial;        // (nchunks, K, Cin, Cout)
  int64_t n_rows;
  int K, Cin, Cout, nchunks, chunk_rows;
  int xcd_order;         // workgroup -> (chunk, offset, tile group) decoded per XCD (k_spconv_wgrad)
  int64_t n_src;         // rows of X, or <= 0 when the caller does not know (then no 32-bit offsets)
  unsigned long long* trace;   // measurement only (sv_debug_wgrad_trace, instance DBG = 16): 8 words per wave, or null
  const float* in_coef;  // (2, Cin) scale | shift: X is read through y = [relu](x * scale + shift) (InNorm), or null
  int in_relu;
};

// The lane's CT input channels are the same in every step (c_base + CT li + c): their coefficients sit in registers, the transform is CT fused
// multiply-adds + CT max per operand load.  Pairs are compacted, so every loaded row is a real neighbour (no validity select); the ring's dummy and
// tail loads are zeroed AFTER the transform or meet a zeroed dY operand.
template <int CT>
struct WgIn {
  float sc[CT], sh[CT], lo;
  bool on;
  __device__ __forceinline__ void init(const float* coef, int relu, int Cin, int c0, bool x_in) {
    on = coef != nullptr;
    lo = relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      sc[c] = (on && x_in) ? coef[c0 + c] : 1.f;
      sh[c] = (on && x_in) ? coef[Cin + c0 + c] : 0.f;
    }
  }
  template <typename XV>
  __device__ __forceinline__ void apply(XV& xs) const {
    if (!on) return;                                         // wave-uniform
    if constexpr (CT == 1) xs = fmaxf(__fmaf_rn(xs, sc[0], sh[0]), lo);
    else {
#pragma unroll
      for (int c = 0; c < CT; ++c) xs[c] = fmaxf(__fmaf_rn(xs[c], sc[c], sh[c]), lo);
    }
  }
};

template <int N> struct WgVec;
template <> struct WgVec<4> { using type = f32x4; };
template <> struct WgVec<2> { using type = f32x2; };
template <> struct WgVec<1> { using type = float; };
__device__ __forceinline__ void wg_gload(f32x4& v, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void wg_gload(f32x2& v, const float* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void wg_gload(float& v, const float* p) { asm volatile("global_load_dword %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
// the same loads addressed as (uniform 64-bit base in SGPRs) + (32-bit byte offset per lane): no 64-bit vector arithmetic per load
__device__ __forceinline__ void wg_gload_s(f32x4& v, uint32_t off, const float* base) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ void wg_gload_s(f32x2& v, uint32_t off, const float* base) { asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(v) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ void wg_gload_s(float& v, uint32_t off, const float* base) { asm volatile("global_load_dword %0, %1, %2" : "=&v"(v) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ float wg_elem(const f32x4& v, int i) { return v[i]; }
__device__ __forceinline__ float wg_elem(const f32x2& v, int i) { return v[i]; }
__device__ __forceinline__ float wg_elem(const float& v, int) { return v; }

// grid = (nchunks, K, tile groups).  Each wave walks its share of the chunk 64 rows at a time: one coalesced read of the
// neighbour table, ballot + prefix popcount compaction of the valid (row, source) pairs into a wave-private LDS list,
// then MFMAs over the COMPACTED pairs only (4 pairs per 16x16x4 step) with the next step's operands requested first.
// The four waves' accumulators are summed through LDS in a fixed order and one slab per (chunk, k) is stored.
// OFF32: operand addresses as 32-bit byte offsets from uniform bases (the launcher checks that every offset fits) and the tail mask only in a
// list's last step -- 35 -> ~15 non-MFMA instructions per 16-MFMA step
template <int CT, int NTL, bool OFF32, int DBG = 0>  // register tile grid: CT x NTL tiles of 16x16 (rows = c_in, cols = c_out); DBG (measurement): 1 no operand loads, 2 no MFMAs
// four waves per SIMD (<4,4>: 122 VGPRs; 140 and three waves unbounded): 4 % slower before the loop was trimmed, 3 % faster on the 139 k-row layer
// and equal elsewhere after it
#ifndef SEEVCN_WGRAD_WAVES
#define SEEVCN_WGRAD_WAVES 4
#endif
__global__ __launch_bounds__(256, SEEVCN_WGRAD_WAVES) void k_spconv_wgrad(WgradArgs a) {
  // the wave's compacted pairs: (source, row) -- OFF32: as byte offsets of the two operand rows, and 32 copies of the last pair behind the list so that
  // the ring's dummy tail loads need no clamp
  __shared__ int2 pjr[4][64 * WG_SUB + 32];
  __shared__ float red[CT * NTL * 256];
  // 1-D grid = (chunk fastest, offset, tile group).  SEEVCN_WGRAD_XCD=1 decodes it instead so that the chunks of one eighth of the rows run on
  // ONE XCD (workgroup b runs on XCD b % 8; a scene's rows then go through one L2 for all 27 offsets, offset-major inside the XCD).  Measured
  // (round 2, 64 -> 64 layers): 170 / 98 us against 155 / 86 us in the plain order -- this kernel is bound by its busiest workgroups (the
  // centre offset has a pair for every row, a corner offset for one row in twenty), not by its 4x over-fetch; the plain order spreads the
  // heavy offsets over all XCDs.  Off by default.
  int k, chunk, zgroup;
  {
    const int cpr = (a.nchunks + 7) / 8;                       // chunks per region
    const int b = blockIdx.x;
    if (a.xcd_order == 2) {                                    // offsets fastest: the 27 offsets of a chunk back to back on its XCD
      const int xcd = b % 8, j = b / 8;
      k = j % a.K;
      chunk = xcd * cpr + (j / a.K) % cpr;
      zgroup = j / (cpr * a.K);
    } else if (a.xcd_order) {
      const int xcd = b % 8, j = b / 8;
      chunk = xcd * cpr + j % cpr;
      k = (j / cpr) % a.K;
      zgroup = j / (cpr * a.K);
    } else {
      chunk = b % a.nchunks;
      k = (b / a.nchunks) % a.K;
      zgroup = b / (a.nchunks * a.K);
    }
    if (chunk >= a.nchunks) return;
  }
  const int ngroups_n = (a.Cout / 16) / NTL;
  const int c_base = (zgroup / ngroups_n) * CT * 16, n_base = (zgroup % ngroups_n) * NTL * 16;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const unsigned long long t_start = (DBG & 16) ? __builtin_amdgcn_s_memtime() : 0ull;     // per-wave stamps: the trace instance only
  unsigned long long t_pro = 0ull, t_loop = 0ull, t_mark = t_start;
  unsigned tr_pairs = 0, tr_passes = 0;
  f32x4 acc[CT][NTL];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < NTL; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t r_begin = (int64_t)chunk * a.chunk_rows, r_end = min(r_begin + (int64_t)a.chunk_rows, a.n_rows);
  const int32_t* nb = a.nbr + (int64_t)k * a.n_rows;

  // Operand fetch for one MFMA step (4 pairs): lane (li, kk) needs X[j_kk][.] for CT tiles and dY[r_kk][.] for NTL tiles.
  // Tile c holds the channels c_base + CT*i + c (i = 0..15), so the CT values of a lane are CONTIGUOUS: one 4*CT-byte load per
  // operand instead of CT scalar loads, and the 16 lanes of a pair read its whole 64*CT-byte row segment.
  // The loads are inline asm: hipcc sinks a plain prefetch load into the block of its first use (measured: load, vmcnt(0), MFMA),
  // an asm load stays where it is written and is retired by the counted s_waitcnt in `consume`.  Steps past the end of the list
  // re-read its last pair (a cache hit); their operands are zeroed after the wait.
  using XV = typename WgVec<CT>::type;
  using YV = typename WgVec<NTL>::type;
  const bool x_in = CT > 1 || c_base + li < a.Cin;           // the 3-channel input layer runs with zero-padded rows
  const float* dYc = a.dY + r_begin * a.Cout;                // this chunk's rows (uniform)
  const uint32_t xconst = (uint32_t)(x_in ? c_base + CT * li : 0) * 4u, yconst = (uint32_t)(n_base + NTL * li) * 4u;
  const uint32_t xrow = (uint32_t)a.Cin * 4u, yrow = (uint32_t)a.Cout * 4u;
  WgIn<CT> win;
  win.init(a.in_coef, a.in_relu, a.Cin, c_base + CT * li, x_in);
  auto issue = [&](int p, int cnt, XV& xs, YV& ys) {
    if constexpr (DBG & 1) {
      xs = XV{} + 1.f, ys = YV{} + 1.f;                        // measurement: what the kernel costs without its operand loads
      asm volatile("" : "+v"(xs), "+v"(ys));
    } else if constexpr (OFF32) {
      const int2 jr = pjr[wid][p];                             // byte offsets; entries past the list repeat its last pair
      wg_gload_s(xs, (uint32_t)jr.x + xconst, a.X);
      wg_gload_s(ys, (uint32_t)jr.y + yconst, dYc);
    } else {
      const int2 jr = pjr[wid][p < cnt ? p : cnt - 1];
      wg_gload(xs, a.X + (int64_t)jr.x * a.Cin + (x_in ? c_base + CT * li : 0));
      wg_gload(ys, a.dY + ((int64_t)r_begin + jr.y) * a.Cout + n_base + NTL * li);
    }
  };
  // waits for the two loads of this step (the 3 younger steps stay in flight), then 16 x CT x NTL MFMAs
  auto consume = [&](int p0, int cnt, XV& xs, YV& ys) {
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(xs), "+v"(ys));
    if (p0 >= cnt) return;                                   // wave-uniform: a dummy step of the ring's tail
    win.apply(xs);
    if constexpr (OFF32) {
      // ONE block of MFMAs (two would get two sets of accumulators); the tail mask is applied in place, and only in a list's last step
      if (p0 + 4 > cnt || !(CT > 1 || c_base + 16 <= a.Cin)) {                       // wave-uniform
        const bool ok = p0 + kk < cnt;
        if (!(ok && x_in)) xs = XV{};
        if (!ok) ys = YV{};
        asm volatile("" : "+v"(xs), "+v"(ys));               // keeps this a BRANCH: if-converted, its selects ran in every step
      }
      if constexpr (DBG & 2) {
        asm volatile("" :: "v"(xs), "v"(ys));                  // measurement: loads and waits only
      } else {
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int t = 0; t < NTL; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wg_elem(xs, c), wg_elem(ys, t), acc[c][t], 0, 0, 0);
      }
    } else {
      const bool ok = p0 + kk < cnt;
      float xa[CT], yb[NTL];
#pragma unroll
      for (int c = 0; c < CT; ++c) xa[c] = (ok && x_in) ? wg_elem(xs, c) : 0.f;
#pragma unroll
      for (int t = 0; t < NTL; ++t) yb[t] = ok ? wg_elem(ys, t) : 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[c], yb[t], acc[c][t], 0, 0, 0);
    }
  };

  // (Tried: the table entries of pass n + 1 requested with asm loads right after the compaction of pass n, waited for at the top of the next
  // iteration -- memory fault: with the destination registers live across the whole MFMA loop hipcc moves them while the loads are in flight.
  // And the pair-list entry of step s + 1 read from LDS one step ahead: no change, 134.3 vs 133.8 us.)
  for (int64_t base = r_begin + wid * (64 * WG_SUB); base < r_end; base += 256 * WG_SUB) {
    // WG_SUB x 64 rows per wave and pass: the neighbour reads are in flight together and the start-up latency of a pass
    // (table read -> compaction -> first operand loads) is paid once per ~80 pairs instead of once per ~20
    int32_t jv[WG_SUB];
#pragma unroll
    for (int s = 0; s < WG_SUB; ++s) {
      const int64_t r = base + s * 64 + lane;
      jv[s] = r < r_end ? nb[r] : -1;
    }
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < WG_SUB; ++s) {
      const unsigned long long m = __ballot(jv[s] >= 0);
      if (jv[s] >= 0) {
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        const int32_t r = (int32_t)(base + s * 64 + lane - r_begin);
        pjr[wid][pos] = OFF32 ? make_int2((int)((uint32_t)jv[s] * xrow), (int)((uint32_t)r * yrow)) : make_int2(jv[s], r);   // one multiply per PAIR, here
      }
      cnt += __popcll(m);
    }
    if (cnt == 0) continue;
    if constexpr (DBG & 8) continue;                         // measurement: table reads and compaction only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-private list: LDS ops of one wave complete in order
    if constexpr (OFF32) {
      if (lane < 32) pjr[wid][cnt + lane] = pjr[wid][cnt - 1];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    // operand ring, 4 steps deep: step s uses stage s % 4 while the loads of steps s+1 .. s+3 are in flight (a gathered row
    // takes ~2 us under load, a step's MFMAs 0.2 us).  EVERY ring slot issues exactly two loads and every consume waits for
    // vmcnt(6): no conditional issue, so each stage register has one definition per slot and hipcc never copies a stage whose
    // load is still in flight (a copied stage lets the late load land in a register that has been handed to something else).
    static_assert(WG_DEPTH == 4, "the wait count in consume() is written for a 4-deep ring");
    if constexpr (DBG & 16) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_pro += t - t_mark, t_mark = t, tr_pairs += (unsigned)cnt, ++tr_passes;
    }
    XV x0, x1, x2, x3;
    YV y0, y1, y2, y3;
    issue(kk, cnt, x0, y0);
    issue(4 + kk, cnt, x1, y1);
    issue(8 + kk, cnt, x2, y2);
    for (int p0 = 0; p0 < cnt; p0 += 16) {
      issue(p0 + 12 + kk, cnt, x3, y3);
      consume(p0, cnt, x0, y0);
      issue(p0 + 16 + kk, cnt, x0, y0);
      consume(p0 + 4, cnt, x1, y1);
      issue(p0 + 20 + kk, cnt, x1, y1);
      consume(p0 + 8, cnt, x2, y2);
      issue(p0 + 24 + kk, cnt, x2, y2);
      consume(p0 + 12, cnt, x3, y3);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2));   // retire the tail's dummy loads
    if constexpr (DBG & 16) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_loop += t - t_mark, t_mark = t;
    }
  }
  const unsigned long long t_body = (DBG & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  if constexpr (DBG & 4) {                                   // measurement: no reduction, no slab store (one value keeps the accumulators alive)
    float sacc = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int t = 0; t < NTL; ++t) sacc += acc[c][t][0] + acc[c][t][1] + acc[c][t][2] + acc[c][t][3];
    if (sacc == 12345.678f) a.partial[0] = sacc;
    return;
  }
  // fixed-order reduction over the 4 waves (wave 0 stores, waves 1..3 add in turn), then one slab per (chunk, k)
  for (int w = 0; w < 4; ++w) {
    if (wid == w) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = &red[((c * NTL + t) * 4 + r) * 64 + lane];
            *d = (w == 0 ? 0.f : *d) + acc[c][t][r];
          }
    }
    __syncthreads();
  }
  // D layout of tile (c,t): col = lane&15, row = 4*(lane>>4) + reg; tile row i is channel c_base + CT*i + c, tile col i is
  // column n_base + NTL*i + t (the interleaved tiles of load_ops)
  float* out = a.partial + (((int64_t)chunk * a.K + k) * a.Cin) * a.Cout;
  for (int e = tid; e < CT * NTL * 256; e += 256) {
    const int ln = e & 63, r = (e >> 6) & 3, tile = e >> 8;
    const int c = tile / NTL, t = tile - c * NTL;
    const int crow = c_base + CT * ((ln >> 4) * 4 + r) + c;
    if (crow < a.Cin) out[(int64_t)crow * a.Cout + n_base + NTL * (ln & 15) + t] = red[e];
  }
  if constexpr (DBG & 16) {
    if (a.trace && lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t_end = __builtin_amdgcn_s_memtime();
      unsigned hw = 0, xcc = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long* o = a.trace + ((size_t)blockIdx.x * 4 + wid) * 8;
      o[0] = t_start, o[1] = t_pro, o[2] = t_loop, o[3] = t_end, o[4] = ((unsigned long long)xcc << 32) | hw, o[5] = t_body,
      o[6] = ((unsigned long long)tr_passes << 32) | tr_pairs, o[7] = ((unsigned long long)k << 32) | (unsigned)chunk;
    }
  }
}

// measurement only: per-wave time stamps of the next MFMA weight-gradient launches (<4,4> instance) go to `buf` (8 x uint64 per wave: 4 per workgroup); null = off
static unsigned long long* g_wgrad_trace = nullptr;
extern "C" int sv_debug_wgrad_trace(void* buf) {
  g_wgrad_trace = static_cast<unsigned long long*>(buf);
  return SV_OK;
}

// generic (any Cin/Cout) stage 1: one thread per (c, n) element, rows of the chunk streamed
__global__ __launch_bounds__(256) void k_spconv_wgrad_valu(WgradArgs a) {
  const int k = blockIdx.y, chunk = blockIdx.x;
  const int64_t r_begin = (int64_t)chunk * a.chunk_rows, r_end = min(r_begin + (int64_t)a.chunk_rows, a.n_rows);
  const int32_t* nb = a.nbr + (int64_t)k * a.n_rows;
  float* out = a.partial + (((int64_t)chunk * a.K + k) * a.Cin) * a.Cout;  // one slab per chunk on this path
  for (int e = threadIdx.x; e < a.Cin * a.Cout; e += blockDim.x) {
    const int c = e / a.Cout, n = e - c * a.Cout;
    float s = 0.f;
    for (int64_t r = r_begin; r < r_end; ++r) {
      const int32_t j = nb[r];
      if (j >= 0) s = fmaf(a.X[(int64_t)j * a.Cin + c], a.dY[r * a.Cout + n], s);
    }
    out[e] = s;
  }
}

// where element (k, c_in, c_out) of the weight gradient goes: contiguous (K, C_in, C_out) or the strides of the caller's parameter layout
// (spconv keeps (C_out, kz, ky, kx, C_in): writing the gradient there directly spares the framework a transposing copy per layer and step)
struct WgradOut {
  int64_t sk, si, so;
  int Cin, Cout, dense;
  __device__ __forceinline__ int64_t at(int64_t e) const {
    if (dense) return e;
    const int co = (int)(e % Cout);
    const int64_t t = e / Cout;
    return (t / Cin) * sk + (t % Cin) * si + co * so;
  }
};

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial, int nchunks, int64_t slab, float* __restrict__ dW, WgradOut o) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < slab; e += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < nchunks; ++c) s += partial[(int64_t)c * slab + e];
    dW[o.at(e)] = s;
  }
}

// slab % 4 == 0: 64 float4 columns x 4 quarters of the chunk range per workgroup -- four times the loads in flight of the scalar
// kernel, partial sums combined in a fixed order ((q0 + q1) + (q2 + q3)): still bitwise reproducible
// Stage 2 reads chunked stage-1 slabs (slab c of nchunks at c * slab4) or the slabs of a stage 1 on equal pieces (k_spconv_wgrad_eq: one (Cin, Cout) slab per
// (piece, group it touches), the slabs of a group back to back).
// quarter q of the four partial sums the reduction kernels make per element: chunk layout -- the q-th quarter of the chunk range; equal pieces -- the
// slabs of the offset's groups in the row eighths 2q and 2q + 1 (runs[2 g], runs[2 g + 1]: first slab and number of slabs of group g = eighth * K + k)
__device__ __forceinline__ f32x4 wgrad_quarter_sum(const float* partial, int nchunks, int64_t slab4, const int32_t* runs, int cc4, int64_t e, int q) {
  const f32x4* p4 = reinterpret_cast<const f32x4*>(partial);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (!runs) {
    const int c0 = (int)((int64_t)nchunks * q / 4), c1 = (int)((int64_t)nchunks * (q + 1) / 4);
    const f32x4* p = p4 + e;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) s += __builtin_nontemporal_load(p + (int64_t)c * slab4);
    return s;
  }
  const int K = (int)(slab4 / cc4), k = (int)(e / cc4);
  const int64_t within = e - (int64_t)k * cc4;
  for (int x = 2 * q; x < 2 * q + 2; ++x) {
    const int g = x * K + k, first = runs[2 * g], n = runs[2 * g + 1];
    const f32x4* p = p4 + (int64_t)first * cc4 + within;
#pragma unroll 4
    for (int c = 0; c < n; ++c) s += __builtin_nontemporal_load(p + (int64_t)c * cc4);
  }
  return s;
}

__global__ __launch_bounds__(256) void k_wgrad_reduce4(const float* __restrict__ partial, int nchunks, int64_t slab4, float* __restrict__ dW, WgradOut o,
                                                       const int32_t* __restrict__ runs) {
  __shared__ f32x4 s_q[4][64];
  const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + col;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (e < slab4) s = wgrad_quarter_sum(partial, nchunks, slab4, runs, o.Cin * o.Cout / 4, e, q);
  s_q[q][col] = s;
  __syncthreads();
  if (q == 0 && e < slab4) {
    const f32x4 v = (s_q[0][col] + s_q[1][col]) + (s_q[2][col] + s_q[3][col]);
    if (o.dense) {
      reinterpret_cast<f32x4*>(dW)[e] = v;
    } else {                                                  // C_out % 4 == 0 here: the four values are consecutive output channels of one (k, c_in)
      const int64_t b = o.at(e * 4);
      dW[b] = v.x, dW[b + o.so] = v.y, dW[b + 2 * o.so] = v.z, dW[b + 3 * o.so] = v.w;
    }
  }
}

extern "C" size_t sv_sparse_conv_wgrad_scratch_bytes(int64_t n_rows, int K, int Cin, int Cout) {
  const int64_t nchunks = (n_rows + 255) / 256;   // worst case: smallest chunk
  return (size_t)(nchunks > 0 ? nchunks : 1) * K * Cin * Cout * sizeof(float);
}

// The launch shape of stage 1 for a layer: register tile grid, tile groups, chunk rows -- a function of the layer's sizes only (the same table gives the
// same slabs and the same summation order in every run)
struct WgradShape {
  bool mfma;
  int tiles_c, tiles_n, groups, chunk_rows, nchunks;
};
static WgradShape wgrad_shape(int64_t n_rows, int K, int Cin, int Cout) {
  WgradShape w{};
  const int ct = (Cin + 15) / 16, nt = Cout / 16;
  w.tiles_c = w.tiles_n = 1;
  // C_in that is not a multiple of 16 (the 3-channel input layer) runs on the MFMA path with zero-padded rows
  w.mfma = Cout % 16 == 0 && (Cin % 16 == 0 || Cin < 16);
  if (w.mfma) {
    if (ct % 4 == 0 && nt % 4 == 0) { w.tiles_c = 4; w.tiles_n = 4; }
    else if (ct % 2 == 0 && nt % 4 == 0) { w.tiles_c = 2; w.tiles_n = 4; }
    else if (ct % 2 == 0 && nt % 2 == 0) { w.tiles_c = 2; w.tiles_n = 2; }
    else if (nt % 2 == 0) { w.tiles_c = 1; w.tiles_n = 2; }
  }
  w.groups = w.mfma ? (ct / w.tiles_c) * (nt / w.tiles_n) : 1;
  w.chunk_rows = wgrad_chunk_rows(n_rows, K, w.groups, Cin);
  w.nchunks = (int)((n_rows + w.chunk_rows - 1) / w.chunk_rows);
  return w;
}

template <int CT, int NTL>
static void launch_wgrad(const WgradArgs& a, hipStream_t st) {
  const int groups = (((a.Cin + 15) / 16) / CT) * ((a.Cout / 16) / NTL);
  const int cpr = (a.nchunks + 7) / 8;
  const unsigned blocks = a.xcd_order ? (unsigned)(8 * cpr * a.K * groups) : (unsigned)(a.nchunks * a.K * groups);
  // 32-bit operand offsets: every source row starts below 2^32 bytes (n_src from the caller); dY offsets are chunk-relative
  static const int off32_env = getenv("SEEVCN_WGRAD_OFF32") ? atoi(getenv("SEEVCN_WGRAD_OFF32")) : 1;
  const bool off32 = off32_env && a.n_src > 0 && (uint64_t)a.n_src * (uint64_t)a.Cin * 4u < 0xffffffffull && (uint64_t)a.chunk_rows * a.Cout * 4u < 0xffffffffull;
  static const int wg_debug = getenv("SEEVCN_WGRAD_DEBUG") ? atoi(getenv("SEEVCN_WGRAD_DEBUG")) : 0;
  // measurement only (gradients are then WRONG): 1 = the narrow layers' stage 1 is not launched at all, 2 = no chunked stage 1 at all -- what the step
  // would gain if these launches were free (its sensitivity to the weight gradients' stream)
  static const int wg_skip = getenv("SEEVCN_WGRAD_SKIP") ? atoi(getenv("SEEVCN_WGRAD_SKIP")) : 0;
  if (wg_skip == 2 || (wg_skip == 1 && !(CT == 4 && NTL == 4))) return;
  if constexpr (CT == 4 && NTL == 4) {
    if (off32 && g_wgrad_trace) {
      WgradArgs t = a;
      t.trace = g_wgrad_trace;
      hipLaunchKernelGGL((k_spconv_wgrad<CT, NTL, true, 16>), dim3(blocks), dim3(256), 0, st, t);
      return;
    }
    if (off32 && wg_debug == 1) { hipLaunchKernelGGL((k_spconv_wgrad<CT, NTL, true, 1>), dim3(blocks), dim3(256), 0, st, a); return; }
    if (off32 && wg_debug == 2) { hipLaunchKernelGGL((k_spconv_wgrad<CT, NTL, true, 2>), dim3(blocks), dim3(256), 0, st, a); return; }
    if (off32 && wg_debug == 4) { hipLaunchKernelGGL((k_spconv_wgrad<CT, NTL, true, 4>), dim3(blocks), dim3(256), 0, st, a); return; }
    if (off32 && wg_debug == 5) { hipLaunchKernelGGL((k_spconv_wgrad<CT, NTL, true, 5>), dim3(blocks), dim3(256), 0, st, a); return; }
    if (off32 && wg_debug == 12) { hipLaunchKernelGGL((k_spconv_wgrad<CT, NTL, true, 12>), dim3(blocks), dim3(256), 0, st, a); return; }
    if (off32 && wg_debug == 8) { hipLaunchKernelGGL((k_spconv_wgrad<CT, NTL, true, 8>), dim3(blocks), dim3(256), 0, st, a); return; }
  }
  if (off32) hipLaunchKernelGGL((k_spconv_wgrad<CT, NTL, true>), dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_spconv_wgrad<CT, NTL, false>), dim3(blocks), dim3(256), 0, st, a);
}

// One pending stage-2 reduction (sv_sparse_conv_wgrad_stage1 -> sv_sparse_conv_wgrad_reduce_batch)
struct WgradReduceJob {
  const float* partial;
  float* dW;
  int64_t slab;          // K * Cin * Cout
  int nslabs, wg0;
  const int32_t* runs;   // slab runs of a stage 1 on equal pieces (wgrad_quarter_sum), or null: nslabs chunk slabs
  WgradOut out;
};

static int wgrad_run(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K,
                                    int Cin, int Cout, void* scratch, void* stream, WgradOut out, WgradReduceJob* defer = nullptr) {
  const InNorm in = take_input_norm();
  SV_CHECK_ARG(n_rows >= 0 && K > 0 && Cin > 0 && Cout > 0 && dW, "sparse_conv_wgrad: bad arguments");
  hipStream_t st = sv_stream(stream);
  const int64_t slab = (int64_t)K * Cin * Cout;
  if (defer) defer->partial = nullptr, defer->nslabs = 0;
  if (n_rows == 0) {
    SV_HIP(hipMemsetAsync(dW, 0, (size_t)slab * 4, st));      // every element, whatever the layout (the strided form is a permutation of the slab)
    return SV_OK;
  }
  SV_CHECK_ARG(X && nbr && dY && scratch, "sparse_conv_wgrad: null pointer");
  const WgradShape w = wgrad_shape(n_rows, K, Cin, Cout);
  // workgroup order: the narrow layers (C_in <= 32: little matrix work per gathered byte) run the chunks of an eighth of the rows on ONE XCD, so that
  // a scene's rows go through one L2 for all 27 offsets -- measured after the loop's instruction stream was trimmed: 16->16 32.7 -> 26.5 us,
  // 16->32 35.1 -> 29.7, 32->32 87.7 -> 74.7, 32->64 56.2 -> 54.0; the 64-channel layers lose with it (135 -> 144 us: they are bound by their
  // busiest workgroups, and the plain order spreads the heavy centre offsets over all XCDs; with the 27 offsets of a chunk back to back on its XCD,
  // order 2, 135 -> 158 us).  SEEVCN_WGRAD_XCD=0/1/2 forces one order for all layers (measurement).
  static const int xcd_env = getenv("SEEVCN_WGRAD_XCD") ? atoi(getenv("SEEVCN_WGRAD_XCD")) : -1;
  const int xcd_order = xcd_env >= 0 ? xcd_env : (Cin <= 32 ? 1 : 0);
  const bool reduce4 = slab % 4 == 0 && Cout % 4 == 0 && (uintptr_t)dW % 16 == 0 && (uintptr_t)scratch % 16 == 0;
  WgradArgs a{X, nbr, dY, reinterpret_cast<float*>(scratch), n_rows, K, Cin, Cout, w.nchunks, w.chunk_rows, xcd_order, n_src};
  a.in_coef = in.coef, a.in_relu = in.relu;
  SV_CHECK_ARG(!in.coef || w.mfma, "sparse_conv_wgrad: an input transform needs an MFMA tile shape (C_in %d, C_out %d)", Cin, Cout);
  const int nslabs = a.nchunks;
  if (w.mfma) {
    if (w.tiles_c == 4) launch_wgrad<4, 4>(a, st);
    else if (w.tiles_c == 2 && w.tiles_n == 4) launch_wgrad<2, 4>(a, st);
    else if (w.tiles_c == 2) launch_wgrad<2, 2>(a, st);
    else if (w.tiles_n == 2) launch_wgrad<1, 2>(a, st);
    else launch_wgrad<1, 1>(a, st);
  } else {
    hipLaunchKernelGGL(k_spconv_wgrad_valu, dim3(a.nchunks, K), dim3(256), 0, st, a);
  }
  if (defer && reduce4) {
    defer->partial = a.partial, defer->dW = dW, defer->slab = slab, defer->nslabs = nslabs, defer->runs = nullptr, defer->out = out;     // summed later, with the other layers' slabs
    SV_LAUNCH_CHECK();
    return SV_OK;
  }
  if (reduce4)
    hipLaunchKernelGGL(k_wgrad_reduce4, dim3(sv_div_up(slab / 4, 64)), dim3(256), 0, st, a.partial, nslabs, slab / 4, dW, out, nullptr);
  else
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(sv_grid_1d(slab, 256)), dim3(256), 0, st, a.partial, nslabs, slab, dW, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_sparse_conv_wgrad(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K,
                                    int Cin, int Cout, void* scratch, void* stream) {
  return wgrad_run(X, n_src, nbr, dY, dW, n_rows, K, Cin, Cout, scratch, stream, WgradOut{0, 0, 0, Cin, Cout, 1});
}

// the same with the gradient written at element strides (stride_k, stride_cin, stride_cout) of dW -- a permutation of the K * C_in * C_out slab
extern "C" int sv_sparse_conv_wgrad_strided(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K,
                                    int Cin, int Cout, int64_t stride_k, int64_t stride_cin, int64_t stride_cout, void* scratch, void* stream) {
  SV_CHECK_ARG(stride_k > 0 && stride_cin > 0 && stride_cout > 0, "sparse_conv_wgrad_strided: strides must be positive");
  return wgrad_run(X, n_src, nbr, dY, dW, n_rows, K, Cin, Cout, scratch, stream, WgradOut{stride_k, stride_cin, stride_cout, Cin, Cout, 0});
}


// ---- stage 2 of SEVERAL layers in one launch: the backward of a backbone runs 12 weight gradients, each followed by a ~6 us reduction launch of its
// own; their results are only needed by the optimiser, so the slabs of every layer can be summed together at the end (same fixed order per element:
// bitwise the same values as k_wgrad_reduce4).
constexpr int WGR_MAX = 16;
struct WgradReduceBatch {
  WgradReduceJob j[WGR_MAX];
  int n;
};
static_assert(sizeof(WgradReduceBatch) <= 3900, "kernel argument block");

__global__ __launch_bounds__(256) void k_wgrad_reduce4_batch(WgradReduceBatch b) {
  __shared__ f32x4 s_q[4][64];
  int ji = 0;
#pragma unroll
  for (int q = 1; q < WGR_MAX; ++q) ji += (q < b.n && (int)blockIdx.x >= b.j[q].wg0) ? 1 : 0;
  const WgradReduceJob& J = b.j[ji];
  const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t slab4 = J.slab / 4, e = (int64_t)((int)blockIdx.x - J.wg0) * 64 + col;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (e < slab4) s = wgrad_quarter_sum(J.partial, J.nslabs, slab4, J.runs, J.out.Cin * J.out.Cout / 4, e, q);
  s_q[q][col] = s;
  __syncthreads();
  if (q == 0 && e < slab4) {
    const f32x4 v = (s_q[0][col] + s_q[1][col]) + (s_q[2][col] + s_q[3][col]);
    if (J.out.dense) {
      reinterpret_cast<f32x4*>(J.dW)[e] = v;
    } else {
      const int64_t o = J.out.at(e * 4);
      J.dW[o] = v.x, J.dW[o + J.out.so] = v.y, J.dW[o + 2 * J.out.so] = v.z, J.dW[o + 3 * J.out.so] = v.w;
    }
  }
}

// bytes of partial slabs stage 1 writes for this layer (exact: the chunking sv_sparse_conv_wgrad will choose), for callers that keep one region per layer
extern "C" size_t sv_sparse_conv_wgrad_partial_bytes(int64_t n_rows, int K, int Cin, int Cout) {
  if (n_rows <= 0 || K <= 0 || Cin <= 0 || Cout <= 0) return 256;
  const WgradShape w = wgrad_shape(n_rows, K, Cin, Cout);
  return ((size_t)w.nchunks * K * Cin * Cout * sizeof(float) + 255) / 256 * 256;
}

// Stage 1 of sv_sparse_conv_wgrad_strided only: the partial slabs go to `partial` (sv_sparse_conv_wgrad_partial_bytes) and *job (10 int64, host) receives
// {partial, dW, slab, nslabs, stride_k, stride_cin, stride_cout, Cin, Cout, slab runs (device address, 0 = chunk slabs)} for sv_sparse_conv_wgrad_reduce_batch; layers stage 2 does not
// take in batch form (odd slab sizes, n_rows = 0) are finished here and leave nslabs = 0.
extern "C" int sv_sparse_conv_wgrad_stage1(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K, int Cin,
                                           int Cout, int64_t stride_k, int64_t stride_cin, int64_t stride_cout, void* partial, int64_t* job, void* stream) {
  SV_CHECK_ARG(job && stride_k > 0 && stride_cin > 0 && stride_cout > 0, "sparse_conv_wgrad_stage1: bad arguments");
  WgradReduceJob d{};
  int rc = wgrad_run(X, n_src, nbr, dY, dW, n_rows, K, Cin, Cout, partial, stream, WgradOut{stride_k, stride_cin, stride_cout, Cin, Cout, 0}, &d);
  job[0] = (int64_t)(uintptr_t)d.partial, job[1] = (int64_t)(uintptr_t)d.dW, job[2] = d.slab, job[3] = d.nslabs;
  job[4] = stride_k, job[5] = stride_cin, job[6] = stride_cout, job[7] = Cin, job[8] = Cout, job[9] = (int64_t)(uintptr_t)d.runs;
  return rc;
}

// jobs_host: n_jobs rows of 10 int64 as written by sv_sparse_conv_wgrad_stage1 (rows with nslabs = 0 are skipped): every layer's slabs summed in one launch
extern "C" int sv_sparse_conv_wgrad_reduce_batch(const int64_t* jobs_host, int n_jobs, void* stream) {
  SV_CHECK_ARG(n_jobs >= 0 && (jobs_host || n_jobs == 0), "sparse_conv_wgrad_reduce_batch: bad arguments");
  hipStream_t st = sv_stream(stream);
  WgradReduceBatch b;
  b.n = 0;
  int wgs = 0;
  for (int q = 0; q < n_jobs; ++q) {
    const int64_t* r = jobs_host + 10 * q;
    if (r[3] <= 0) continue;
    SV_CHECK_ARG(r[0] && r[1] && r[2] > 0 && r[2] % 4 == 0, "sparse_conv_wgrad_reduce_batch: job %d: bad slab", q);
    WgradReduceJob& J = b.j[b.n];
    J.partial = reinterpret_cast<const float*>((uintptr_t)r[0]), J.dW = reinterpret_cast<float*>((uintptr_t)r[1]), J.slab = r[2], J.nslabs = (int)r[3];
    J.out = WgradOut{r[4], r[5], r[6], (int)r[7], (int)r[8], 0};
    J.runs = reinterpret_cast<const int32_t*>((uintptr_t)r[9]);
    J.wg0 = wgs;
    wgs += sv_div_up(r[2] / 4, 64);
    if (++b.n == WGR_MAX) {
      hipLaunchKernelGGL(k_wgrad_reduce4_batch, dim3(wgs), dim3(256), 0, st, b);
      b.n = 0, wgs = 0;
    }
  }
  if (b.n > 0) hipLaunchKernelGGL(k_wgrad_reduce4_batch, dim3(wgs), dim3(256), 0, st, b);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient on EQUAL PIECES.
// What tools/wgrad_trace.py measured on the chunked kernel above (64 -> 64 at 139 k rows, 134 us, 76 TFLOP/s): the busiest SIMD needs 0.77-0.83 of
// the launch's span for its MFMAs alone while the average SIMD needs 0.54 -- pairs per SIMD max / mean = 1.4-1.6.  A (chunk, offset) workgroup's
// work follows the offset's density (centre offset: a pair for every row; out-of-plane corners: one row in nine), the dispatcher places workgroups
// by free slots, not by work, and a workgroup is 40 % of a CU's fair share: the launch ends when its unluckiest CU does.  Dispatch order (heavy
// offsets first) and half chunks for the heavy offsets were tried and changed nothing (profiles/r04_wgrad_trace.txt).
// Here the work is cut to fit the machine instead: the table's pairs, in offset-major order, are cut into exactly as many pieces as workgroups are
// resident (4 or 8 per CU), each piece the same number of pairs up to one 64-row unit.  A piece is a run of 64-row units; it touches one to three
// offsets and writes one (Cin, Cout) slab per offset it touches (fewer slabs than the chunked form: pieces + K against chunks x K); inside a piece
// the four waves take pair-exact quarters, so a SIMD's share is even too.  The cuts come from a per-table plan (sv_wgrad_plan_build: unit pair
// counts, their prefix sums, the cuts, the first slab of every piece, the slab run of every offset), a function of the table alone: same table,
// same slabs, same summation order -- bitwise reproducible like the chunked form (the VALUES differ from the chunked form's in the last bits: other
// partial sums).
// ------------------------------------------------------------------------------------------------
constexpr int WGE_CUS = 256;              // MI355X
constexpr int WGE_UNIT = 64;              // rows per unit
constexpr int WGE_EIGHTHS = 8;            // row eighths of the unit order: one per XCD
constexpr int WGE_MAX_PIECES = WGE_CUS * 8;

// plan layout (int32; every part padded to a multiple of 4): cut[pieces + 1] | slab0[pieces + 1] | runs[2 * 8 K] | prefix[U + 1] (pairs in front of every 64-row unit in (eighth, offset, unit) order; U = 8 K * units per group)
struct WgradPlanPtrs {
  int32_t *cut, *slab0, *runs, *pre;
};
static int wgp_pad4(int n) { return (n + 3) & ~3; }
static WgradPlanPtrs wgrad_plan_ptrs(void* plan, int pieces, int K) {
  WgradPlanPtrs p;
  p.cut = static_cast<int32_t*>(plan);
  p.slab0 = p.cut + wgp_pad4(pieces + 1);
  p.runs = p.slab0 + wgp_pad4(pieces + 1);
  p.pre = p.runs + wgp_pad4(2 * WGE_EIGHTHS * K);
  return p;
}
// Unit order: (row eighth x, offset k, unit j inside the eighth) -- group g = x * K + k holds the WGE-unit slots of one offset inside one eighth of the
// rows; unit u of group g is rows [64 r, 64 r + 64), r = x * nbu8 + (u - g * nbu8).  Pieces are cut along this order and workgroup b takes piece
// (b % 8) * (pieces / 8) + b / 8: the pieces of XCD x (workgroup b runs on XCD b % 8) lie in the x-th eighth of the order, i.e. (up to the drift of
// the cuts) in the x-th eighth of the ROWS -- a row's X / dY go through one L2 for all 27 offsets instead of through eight.  (Offset-major over the
// whole table, the first version, pulled 4.7x the algorithmic bytes: set d of the round-4 profiles, in git history; now 2.2x, profiles/r04_e_traffic.json.)
static int64_t wgrad_units_per_group(int64_t n_rows) {
  const int64_t nbu = (n_rows + WGE_UNIT - 1) / WGE_UNIT;
  return (nbu + WGE_EIGHTHS - 1) / WGE_EIGHTHS;
}

extern "C" size_t sv_wgrad_plan_bytes(int64_t n_rows, int K, int pieces) {
  if (n_rows < 0 || K <= 0 || pieces <= 0) return 0;
  const int64_t U = wgrad_units_per_group(n_rows) * WGE_EIGHTHS * K;
  return ((size_t)(2 * wgp_pad4(pieces + 1) + wgp_pad4(2 * WGE_EIGHTHS * K) + U + 1) * sizeof(int32_t) + 255) / 256 * 256;
}

// pieces the kernel instance of a layer shape is cut for: four workgroups per CU for the 64-channel-multiple layers (122 VGPRs: four waves per
// SIMD), eight for the narrower instances (their workgroups are short and latency-bound: the chunked form also ran them on twice the workgroups)
extern "C" int sv_wgrad_plan_pieces(int Cin, int Cout) {
  const int ct = (Cin + 15) / 16, nt = Cout / 16;
  static const int per_cu = getenv("SEEVCN_WGRAD_PIECES_PER_CU") ? atoi(getenv("SEEVCN_WGRAD_PIECES_PER_CU")) : 0;      // measurement switch (1..8)
  if (per_cu >= 1 && per_cu <= 8) return WGE_CUS * per_cu;
  return (ct % 4 == 0 && nt % 4 == 0) ? WGE_CUS * 4 : WGE_CUS * 8;
}

// one workgroup per table: counts -> exclusive prefix (in place; pre[U] = all pairs), the cuts, the first slab of every piece, the slab run of every offset.
// (The first version walked a thread's ~58 counts with one dependent load per iteration: 47-98 us per table.  Here a thread's run is a whole number of
// int4, loaded four at a time.)
struct WgradPlanJob {
  const int32_t* nbr;
  int64_t n_rows;
  int32_t *pre, *cut, *slab0, *runs;
  int U, nbu, K, pieces, wg0;          // nbu: unit slots per group (wgrad_units_per_group); wg0: first workgroup of this table in the batched count launch
};
constexpr int WGP_MAX = 12;
struct WgradPlanBatch {
  WgradPlanJob j[WGP_MAX];
  int n;
};
static_assert(sizeof(WgradPlanBatch) <= 3900, "kernel argument block");

// pairs of every 64-row unit of every table.  A wave reads 256 consecutive rows of one offset with one int4 per lane (lane l: rows 4 l .. 4 l + 3, so
// unit j of the four is lanes 16 j .. 16 j + 15) and counts each unit from the four component ballots; a workgroup = 4 waves x WGP_LOADS such loads.
// (First version: one 4-byte load per lane and one unit per wave, 29 k workgroups for the three tables of a step: 32 us.)
constexpr int WGP_LOADS = 4;
constexpr int WGP_ROWS = 4 * 256 * WGP_LOADS;     // rows of one offset per workgroup
__global__ __launch_bounds__(256) void k_wgrad_plan_count(WgradPlanBatch b) {
  int ji = 0;
#pragma unroll
  for (int q = 1; q < WGP_MAX; ++q) ji += (q < b.n && (int)blockIdx.x >= b.j[q].wg0) ? 1 : 0;
  const WgradPlanJob& J = b.j[ji];
  const int64_t rows_cov = (int64_t)J.nbu * WGE_EIGHTHS * WGE_UNIT;      // every unit slot of every group, the empty ones past the table's end included
  const int wgs_per_k = (int)((rows_cov + WGP_ROWS - 1) / WGP_ROWS), local = (int)blockIdx.x - J.wg0;
  const int k = local / wgs_per_k, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int32_t* __restrict__ nb = J.nbr + (int64_t)k * J.n_rows;
  const bool aligned = ((uintptr_t)nb & 15) == 0;                       // n_rows % 4 != 0 shifts the offsets' rows off 16 bytes
  const int64_t r0 = (int64_t)(local % wgs_per_k) * WGP_ROWS + (int64_t)wid * (256 * WGP_LOADS);
  int4 v[WGP_LOADS];
#pragma unroll
  for (int q = 0; q < WGP_LOADS; ++q) {
    const int64_t r = r0 + q * 256 + 4 * lane;
    if (aligned && r + 3 < J.n_rows) v[q] = *reinterpret_cast<const int4*>(nb + r);
    else {
      v[q].x = r < J.n_rows ? nb[r] : -1, v[q].y = r + 1 < J.n_rows ? nb[r + 1] : -1;
      v[q].z = r + 2 < J.n_rows ? nb[r + 2] : -1, v[q].w = r + 3 < J.n_rows ? nb[r + 3] : -1;
    }
  }
#pragma unroll
  for (int q = 0; q < WGP_LOADS; ++q) {
    const unsigned long long m0 = __ballot(v[q].x >= 0), m1 = __ballot(v[q].y >= 0), m2 = __ballot(v[q].z >= 0), m3 = __ballot(v[q].w >= 0);
    if (lane < 4) {
      const unsigned long long mask = 0xffffull << (16 * lane);
      const int64_t unit = (r0 + q * 256) / WGE_UNIT + lane;          // row unit of the table
      if (unit < (int64_t)J.nbu * WGE_EIGHTHS) {
        const int x = (int)(unit / J.nbu), j = (int)(unit % J.nbu);
        J.pre[((int64_t)x * J.K + k) * J.nbu + j] = __popcll(m0 & mask) + __popcll(m1 & mask) + __popcll(m2 & mask) + __popcll(m3 & mask);
      }
    }
  }
}

// 1024 threads: inclusive scan of one value per thread -- shuffles inside a wave, the 16 wave totals through LDS (two barriers)
__device__ __forceinline__ int wgp_block_scan_inclusive(int* s_wave, int v, int tid) {
  const int lane = tid & 63, wid = tid >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  __syncthreads();                                       // s_wave may still be read from the previous scan
  if (lane == 63) s_wave[wid] = v;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) base += w < wid ? s_wave[w] : 0;
  return v + base;
}

__global__ __launch_bounds__(1024) void k_wgrad_plan_cuts(WgradPlanBatch b) {
  __shared__ int s_wave[16];
  __shared__ int s_run[1025];                            // pairs in front of every thread's run of units (+ the total): the coarse level of the cut search
  const WgradPlanJob& J = b.j[blockIdx.x];
  int32_t* __restrict__ pre = J.pre;
  const int U = J.U, nbu = J.nbu, K = J.K * WGE_EIGHTHS, pieces = J.pieces;      // K here: GROUPS (eighth, offset)
  const int tid = threadIdx.x;
  // a thread's run: `per` counts, a multiple of 16, so that it is whole groups of four int4 (pre is 16-byte aligned: the plan's parts are multiples
  // of 4 ints); only the table's last run has a remainder
  const int per = (((U + 1023) / 1024) + 15) & ~15, b0 = min(U, tid * per), b1 = min(U, b0 + per);
  // runs of at most 64 counts (tables up to 65 k units: every table of the benchmarked step) stay in registers between the sum and the write-back:
  // sixteen int4 loads in flight once, instead of two passes of four dependent rounds
  const bool in_regs = per <= 64;
  int4 keep[16];
  int sum = 0;
  if (in_regs) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int u = b0 + 4 * q;
      if (u + 4 <= b1) keep[q] = *reinterpret_cast<const int4*>(pre + u);
      else keep[q] = make_int4(u < b1 ? pre[u] : 0, u + 1 < b1 ? pre[u + 1] : 0, u + 2 < b1 ? pre[u + 2] : 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) sum += (keep[q].x + keep[q].y) + (keep[q].z + keep[q].w);
  } else {
    int u = b0;
    for (; u + 16 <= b1; u += 16) {
      const int4 v0 = *reinterpret_cast<const int4*>(pre + u), v1 = *reinterpret_cast<const int4*>(pre + u + 4);
      const int4 v2 = *reinterpret_cast<const int4*>(pre + u + 8), v3 = *reinterpret_cast<const int4*>(pre + u + 12);
      sum += (v0.x + v0.y + v0.z + v0.w) + (v1.x + v1.y + v1.z + v1.w) + (v2.x + v2.y + v2.z + v2.w) + (v3.x + v3.y + v3.z + v3.w);
    }
    for (; u < b1; ++u) sum += pre[u];
  }
  const int incl = wgp_block_scan_inclusive(s_wave, sum, tid);
  s_run[tid] = incl - sum;
  if (tid == 1023) s_run[1024] = incl;
  if (in_regs) {
    int run = incl - sum;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int u = b0 + 4 * q;
      int4 o;
      o.x = run, run += keep[q].x;
      o.y = run, run += keep[q].y;
      o.z = run, run += keep[q].z;
      o.w = run, run += keep[q].w;
      if (u + 4 <= b1) *reinterpret_cast<int4*>(pre + u) = o;
      else {
        if (u < b1) pre[u] = o.x;
        if (u + 1 < b1) pre[u + 1] = o.y;
        if (u + 2 < b1) pre[u + 2] = o.z;
      }
    }
  } else {
    int run = incl - sum, u = b0;
    for (; u + 16 <= b1; u += 16) {
      int4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const int4*>(pre + u + 4 * q);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int4 o;
        o.x = run, run += v[q].x;
        o.y = run, run += v[q].y;
        o.z = run, run += v[q].z;
        o.w = run, run += v[q].w;
        *reinterpret_cast<int4*>(pre + u + 4 * q) = o;
      }
    }
    for (; u < b1; ++u) {
      const int c = pre[u];
      pre[u] = run;
      run += c;
    }
  }
  __threadfence_block();
  __syncthreads();
  const long long total = s_run[1024];
  if (tid == 0) pre[U] = (int)total;
  int32_t* __restrict__ cut = J.cut;
  // cut[i] = first unit whose pairs-before reach i * total / pieces (a unit belongs to the piece its FIRST pair falls into; empty units go with a
  // neighbour); pieces may be empty (a table with fewer units than pieces).  Two levels: the run in LDS, then the unit inside the run.
  for (int i = tid; i <= pieces; i += 1024) {
    const long long target = total * i / pieces;
    int lo;
    if (i == pieces) lo = U;
    else if (i == 0) lo = 0;
    else {
      // last run t whose first unit has fewer than `target` pairs in front of it: the answer lies in (t * per, (t + 1) * per]
      int a = 0, c = 1023;
      while (a < c) {
        const int mid = (a + c + 1) >> 1;
        if (s_run[mid] < target && mid * per < U) a = mid;
        else c = mid - 1;
      }
      lo = min(U, a * per);
      int hi = min(U, (a + 1) * per);                   // pre[hi] >= target (the next run's first unit, or the total)
      while (lo < hi) {                                  // smallest u in [lo, hi] with pre[u] >= target
        const int mid = (lo + hi) >> 1;
        const int pm = mid == U ? (int)total : pre[mid];
        if (pm >= target) hi = mid;
        else lo = mid + 1;
      }
    }
    cut[i] = lo;
  }
  __threadfence_block();
  __syncthreads();
  // slabs: one per (piece, offset it touches), numbered in piece order
  static_assert(WGE_MAX_PIECES <= 2048, "two pieces per thread");
  int seg[2] = {0, 0}, first[2];
  for (int q = 0; q < 2; ++q) {
    const int i = tid + q * 1024;
    if (i < pieces) {
      const int u0 = cut[i], u1 = cut[i + 1];
      seg[q] = u0 < u1 ? (u1 - 1) / nbu - u0 / nbu + 1 : 0;
    }
  }
  int carry = 0;
  for (int q = 0; q < 2; ++q) {
    const int inc = wgp_block_scan_inclusive(s_wave, seg[q], tid);
    first[q] = carry + inc - seg[q];
    __syncthreads();
    if (tid == 1023) s_run[0] = inc;
    __syncthreads();
    carry += s_run[0];
  }
  int32_t* __restrict__ slab0 = J.slab0;
  int32_t* __restrict__ runs = J.runs;
  for (int q = 0; q < 2; ++q) {
    const int i = tid + q * 1024;
    if (i < pieces) {
      slab0[i] = first[q];
      if (i == pieces - 1) slab0[pieces] = first[q] + seg[q];
      const int u0 = cut[i], u1 = cut[i + 1];
      if (u0 < u1) {
        const int kf = u0 / nbu, kl = (u1 - 1) / nbu;
        for (int k = kf; k <= kl; ++k) {
          if (u0 <= k * nbu) runs[2 * k] = first[q] + (k - kf);                     // holds the offset's first unit: its run starts here
          if (u1 >= (k + 1) * nbu) runs[2 * k + 1] = first[q] + (k - kf);           // holds its last unit: the run's last slab (turned into a count below)
        }
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int k = tid; k < K; k += 1024) runs[2 * k + 1] = runs[2 * k + 1] - runs[2 * k] + 1;
}

static int wgrad_plan_job(WgradPlanJob& J, const int32_t* nbr, int64_t n_rows, int K, int pieces, void* plan, const char* who) {
  SV_CHECK_ARG(n_rows >= 1 && K >= 1 && K <= 1024 && plan && nbr, "%s: bad arguments", who);
  SV_CHECK_ARG(pieces >= 1 && pieces <= WGE_MAX_PIECES, "%s: 1..%d pieces", who, WGE_MAX_PIECES);
  SV_CHECK_ARG((uintptr_t)plan % 16 == 0, "%s: the plan must be 16-byte aligned", who);
  const int64_t nbu = wgrad_units_per_group(n_rows), U = nbu * WGE_EIGHTHS * K;
  SV_CHECK_ARG(pieces % WGE_EIGHTHS == 0, "%s: the piece count must be a multiple of %d", who, WGE_EIGHTHS);
  SV_CHECK_ARG(U < (1ll << 30) && n_rows * (int64_t)K < (1ll << 31), "%s: table too large for 32-bit unit indices / pair counts", who);
  const WgradPlanPtrs p = wgrad_plan_ptrs(plan, pieces, K);
  J.nbr = nbr, J.n_rows = n_rows, J.pre = p.pre, J.cut = p.cut, J.slab0 = p.slab0, J.runs = p.runs;
  J.U = (int)U, J.nbu = (int)nbu, J.K = K, J.pieces = pieces, J.wg0 = 0;
  return SV_OK;
}
static void wgrad_plan_launch(WgradPlanBatch& b, int wgs, hipStream_t st) {
  hipLaunchKernelGGL(k_wgrad_plan_count, dim3(wgs), dim3(256), 0, st, b);
  hipLaunchKernelGGL(k_wgrad_plan_cuts, dim3(b.n), dim3(1024), 0, st, b);
}

extern "C" int sv_wgrad_plan_build(const int32_t* nbr, int64_t n_rows, int K, int pieces, void* plan, void* stream) {
  WgradPlanBatch b;
  b.n = 1;
  if (int rc = wgrad_plan_job(b.j[0], nbr, n_rows, K, pieces, plan, "sv_wgrad_plan_build")) return rc;
  wgrad_plan_launch(b, (int)(((int64_t)b.j[0].nbu * WGE_EIGHTHS * WGE_UNIT + WGP_ROWS - 1) / WGP_ROWS) * K, sv_stream(stream));
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// the plans of several tables in two launches; jobs_host: n_jobs rows of 8 int64 = {nbr, n_rows, K, pieces, plan, 0, 0, 0} (device addresses)
extern "C" int sv_wgrad_plan_build_batch(const int64_t* jobs_host, int n_jobs, void* stream) {
  SV_CHECK_ARG(n_jobs >= 0 && (jobs_host || n_jobs == 0), "sv_wgrad_plan_build_batch: bad arguments");
  hipStream_t st = sv_stream(stream);
  WgradPlanBatch b;
  b.n = 0;
  int wgs = 0;
  for (int q = 0; q < n_jobs; ++q) {
    const int64_t* r = jobs_host + 8 * q;
    WgradPlanJob& J = b.j[b.n];
    if (int rc = wgrad_plan_job(J, reinterpret_cast<const int32_t*>((uintptr_t)r[0]), r[1], (int)r[2], (int)r[3], reinterpret_cast<void*>((uintptr_t)r[4]),
                                "sv_wgrad_plan_build_batch"))
      return rc;
    J.wg0 = wgs;
    wgs += (int)(((int64_t)J.nbu * WGE_EIGHTHS * WGE_UNIT + WGP_ROWS - 1) / WGP_ROWS) * J.K;
    if (++b.n == WGP_MAX) {
      wgrad_plan_launch(b, wgs, st);
      b.n = 0, wgs = 0;
    }
  }
  if (b.n > 0) wgrad_plan_launch(b, wgs, st);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

struct WgradPlanView {
  const int32_t* cut;
  const int32_t* slab0;
  const int32_t* pre;       // pairs in front of every unit (U + 1)
  int pieces, nbu;
};

// One workgroup per piece.  Per offset the piece touches (a segment): the four waves compact the segment's pairs into ONE list in LDS -- each wave a
// contiguous quarter of the units, written at the position the plan's prefix sums give (so the list is in row order and no counts are exchanged) --
// and then take PAIR-EXACT quarters of the list through the operand ring of the chunked kernel.  Units are 64 rows: dealt to the waves unit by unit a
// centre-offset piece (19 units) gave its waves 5 / 5 / 5 / 4 units, the fourth SIMD of every CU 20 % less work and every workgroup a wait at its
// reduction (pairs per SIMD max / mean 1.13); and an out-of-plane offset (7 pairs per unit) started an MFMA loop per ~28 pairs (30-50 % of such a
// piece's life in loop start-ups).  Here a wave starts ONE loop per segment, over a quarter of its pairs up to one 4-pair step.
// A segment longer than the list (tables with more than WGE_LIST_CAP pairs per piece) runs in several chunks, cut at units by the prefix sums.
constexpr int WGE_LIST_CAP = 1536;         // pairs per chunk: 12 KB
constexpr int WGE_READ = 8;                // units whose table entries a wave requests at once

template <int CT, int NTL, int DBG = 0>
__global__ __launch_bounds__(256, SEEVCN_WGRAD_WAVES) void k_spconv_wgrad_eq(WgradArgs a, WgradPlanView pl) {
  __shared__ int2 plist[WGE_LIST_CAP + 32];
  __shared__ float red[CT * NTL * 256];
  __shared__ int s_chunk[2];
  const int bl = blockIdx.x % pl.pieces, zgroup = blockIdx.x / pl.pieces;
  const int piece = (bl % WGE_EIGHTHS) * (pl.pieces / WGE_EIGHTHS) + bl / WGE_EIGHTHS;      // XCD bl % 8 works in the bl % 8-th eighth of the unit order
  const int u0 = pl.cut[piece], u1 = pl.cut[piece + 1];
  if (u0 >= u1) return;
  int slab = pl.slab0[piece];
  const int ngroups_n = (a.Cout / 16) / NTL;
  const int c_base = (zgroup / ngroups_n) * CT * 16, n_base = (zgroup % ngroups_n) * NTL * 16;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const unsigned long long t_start = (DBG & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned long long t_pro = 0ull, t_loop = 0ull, t_mark = t_start, t_tail = 0ull;
  unsigned tr_pairs = 0, tr_passes = 0;
  using XV = typename WgVec<CT>::type;
  using YV = typename WgVec<NTL>::type;
  const bool x_in = CT > 1 || c_base + li < a.Cin;
  const uint32_t xconst = (uint32_t)(x_in ? c_base + CT * li : 0) * 4u, yconst = (uint32_t)(n_base + NTL * li) * 4u;
  const uint32_t xrow = (uint32_t)a.Cin * 4u, yrow = (uint32_t)a.Cout * 4u;
  f32x4 acc[CT][NTL];
  WgIn<CT> win;
  win.init(a.in_coef, a.in_relu, a.Cin, c_base + CT * li, x_in);
  auto issue = [&](int p, XV& xs, YV& ys) {
    const int2 jr = plist[p];                                // byte offsets; entries past the wave's share are other pairs or the padding (row 0)
    wg_gload_s(xs, (uint32_t)jr.x + xconst, a.X);
    wg_gload_s(ys, (uint32_t)jr.y + yconst, a.dY);
  };
  auto consume = [&](int p0, int pend, XV& xs, YV& ys) {
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(xs), "+v"(ys));
    if (p0 >= pend) return;                                  // wave-uniform: a dummy step of the ring's tail
    win.apply(xs);
    if (p0 + 4 > pend || !(CT > 1 || c_base + 16 <= a.Cin)) {                      // wave-uniform: the tail mask only in a share's last step
      const bool ok = p0 + kk < pend;
      if (!(ok && x_in)) xs = XV{};
      if (!ok) ys = YV{};
      asm volatile("" : "+v"(xs), "+v"(ys));                 // keeps this a BRANCH: if-converted, its selects ran in every step
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int t = 0; t < NTL; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wg_elem(xs, c), wg_elem(ys, t), acc[c][t], 0, 0, 0);
  };

  for (int us = u0; us < u1;) {                              // one segment per (row eighth, offset) group the piece touches
    const int grp = us / pl.nbu, ue = min(u1, (grp + 1) * pl.nbu);        // group = (row eighth, offset)
    const int k = grp % a.K, ru0 = (grp / a.K) * pl.nbu - grp * pl.nbu;    // row unit of unit u: u + ru0
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int t = 0; t < NTL; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int32_t* nb = a.nbr + (int64_t)k * a.n_rows;
    for (int uc = us; uc < ue;) {                            // chunks of at most WGE_LIST_CAP pairs (normally one: a piece is total / pieces pairs)
      if (tid == 0) {
        const int p_first = pl.pre[uc];
        int nu = ue - uc;
        if (pl.pre[ue] - p_first > WGE_LIST_CAP) {           // largest run of units that fits (a unit is at most 64 pairs: at least 24 units)
          int lo = 1, hi = nu;
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pl.pre[uc + mid] - p_first <= WGE_LIST_CAP) lo = mid;
            else hi = mid - 1;
          }
          nu = lo;
        }
        s_chunk[0] = nu, s_chunk[1] = pl.pre[uc + nu] - p_first;
      }
      __syncthreads();                                       // also: every wave is done with the previous chunk's list
      const int nu = s_chunk[0], n = s_chunk[1];
      // the short compaction goes in front of the other waves' MFMA streams: the fourth workgroup of a CU (youngest waves) spent 100 k cycles in it behind
      // three older waves' loops, the first 40 k (tools/wgrad_trace.py); 108.1 -> 106.6 us at 139 k rows.  SEEVCN_WGRAD_PRIO=0: off (A/B)
      if (a.xcd_order) __builtin_amdgcn_s_setprio(3);
      // compaction: wave w takes the units [uc + w nu / 4, uc + (w + 1) nu / 4) and writes their pairs where the prefix sums put them
      {
        const int ua = uc + (int)((int64_t)nu * wid / 4), ub = uc + (int)((int64_t)nu * (wid + 1) / 4);
        int pos0 = ua < ub ? pl.pre[ua] - pl.pre[uc] : 0;
        for (int u = ua; u < ub; u += WGE_READ) {
          int32_t jv[WGE_READ];
#pragma unroll
          for (int s = 0; s < WGE_READ; ++s) {
            const int64_t r = (int64_t)(u + s + ru0) * WGE_UNIT + lane;
            jv[s] = (u + s < ub && r < a.n_rows) ? nb[r] : -1;
          }
#pragma unroll
          for (int s = 0; s < WGE_READ; ++s) {
            const unsigned long long m = __ballot(jv[s] >= 0);
            if (jv[s] >= 0) {
              const uint32_t r = (uint32_t)((u + s + ru0) * WGE_UNIT + lane);
              plist[pos0 + __popcll(m & ((1ull << lane) - 1ull))] = make_int2((int)((uint32_t)jv[s] * xrow), (int)(r * yrow));
            }
            pos0 += __popcll(m);
          }
        }
        if (wid == 0 && lane < 32) plist[n + lane] = make_int2(0, 0);     // what the ring's tail loads of the last share read: row 0, masked
      }
      __syncthreads();
      if constexpr (DBG & 16) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        t_pro += t - t_mark, t_mark = t, ++tr_passes;
      }
      if (a.xcd_order) __builtin_amdgcn_s_setprio(0);
      // pair-exact shares, whole 4-pair steps: wave w takes the pairs [w q, min(n, (w + 1) q))
      const int q = ((n + 15) >> 4) << 2, pb = wid * q, pend = min(n, pb + q);
      if (pb < pend) {
        if constexpr (DBG & 16) tr_pairs += (unsigned)(pend - pb);
        XV x0, x1, x2, x3;
        YV y0, y1, y2, y3;
        issue(pb + kk, x0, y0);
        issue(pb + 4 + kk, x1, y1);
        issue(pb + 8 + kk, x2, y2);
        for (int p0 = pb; p0 < pend; p0 += 16) {
          issue(p0 + 12 + kk, x3, y3);
          consume(p0, pend, x0, y0);
          issue(p0 + 16 + kk, x0, y0);
          consume(p0 + 4, pend, x1, y1);
          issue(p0 + 20 + kk, x1, y1);
          consume(p0 + 8, pend, x2, y2);
          issue(p0 + 24 + kk, x2, y2);
          consume(p0 + 12, pend, x3, y3);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2));   // retire the tail's dummy loads
      }
      if constexpr (DBG & 16) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        t_loop += t - t_mark, t_mark = t;
      }
      uc += nu;
    }
    // fixed-order reduction over the 4 waves (wave 0 stores, waves 1..3 add in turn), then the segment's slab
    for (int w = 0; w < 4; ++w) {
      if (wid == w) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* d = &red[((c * NTL + t) * 4 + r) * 64 + lane];
              *d = (w == 0 ? 0.f : *d) + acc[c][t][r];
            }
      }
      __syncthreads();
    }
    float* out = a.partial + ((int64_t)slab * a.Cin) * a.Cout;
    for (int e = tid; e < CT * NTL * 256; e += 256) {
      const int ln = e & 63, r = (e >> 6) & 3, tile = e >> 8;
      const int c = tile / NTL, t = tile - c * NTL;
      const int crow = c_base + CT * ((ln >> 4) * 4 + r) + c;
      if (crow < a.Cin) out[(int64_t)crow * a.Cout + n_base + NTL * (ln & 15) + t] = red[e];
    }
    ++slab, us = ue;
    __syncthreads();                                         // `red` is rewritten by the next segment's reduction
    if constexpr (DBG & 16) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_tail += t - t_mark, t_mark = t;
    }
  }
  if constexpr (DBG & 16) {
    if (a.trace && lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t_end = __builtin_amdgcn_s_memtime();
      unsigned hw = 0, xcc = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long* o = a.trace + ((size_t)blockIdx.x * 4 + wid) * 8;
      // word 5: t_end - (time behind the last loop of every segment), so that tools/wgrad_trace.py's "after the last pass" is the sum over segments
      o[0] = t_start, o[1] = t_pro, o[2] = t_loop, o[3] = t_end, o[4] = ((unsigned long long)xcc << 32) | hw, o[5] = t_end - t_tail,
      o[6] = ((unsigned long long)tr_passes << 32) | tr_pairs, o[7] = ((unsigned long long)((u0 / pl.nbu) % a.K) << 32) | (unsigned)piece;
    }
  }
}

template <int CT, int NTL>
static void launch_wgrad_eq(const WgradArgs& a, const WgradPlanView& pl, hipStream_t st) {
  const int groups = (((a.Cin + 15) / 16) / CT) * ((a.Cout / 16) / NTL);
  const unsigned blocks = (unsigned)(pl.pieces * groups);
  if constexpr (CT == 4 && NTL == 4) {
    if (g_wgrad_trace) {
      WgradArgs t = a;
      t.trace = g_wgrad_trace;
      hipLaunchKernelGGL((k_spconv_wgrad_eq<CT, NTL, 16>), dim3(blocks), dim3(256), 0, st, t, pl);
      return;
    }
  }
  hipLaunchKernelGGL((k_spconv_wgrad_eq<CT, NTL>), dim3(blocks), dim3(256), 0, st, a, pl);
}

// 1 iff the equal-pieces kernel takes this layer: an MFMA tile shape, operand rows addressable with 32-bit byte offsets
// -- AND it pays: the 64-channel-multiple layers (MFMA-bound: 133 -> 121 us at 139 k rows) with enough table to give the chunked form its ~1000
// workgroups.  The narrow layers are bound by their gathers, not by the matrix pipe: offset-major pieces take their rows through eight L2s 27 times
// (16 -> 16 at 240 k rows: 27 -> 79 us), and a small table pays for pieces + K slabs it does not need (64 -> 128, K = 3, 59 k rows: 42 -> 57 us).
extern "C" int sv_wgrad_planned_applies(int64_t n_src, int64_t n_rows, int K, int Cin, int Cout) {
  const WgradShape w = wgrad_shape(n_rows > 0 ? n_rows : 1, K, Cin, Cout);
  if (!w.mfma || n_rows < 1 || n_src < 1) return 0;
  const char* fe = getenv("SEEVCN_WGRAD_PLANNED_ALL");       // tests: every MFMA shape and size (read per call: a test sets it for itself)
  const int force = fe ? atoi(fe) : 0;
  if (!force && !(w.tiles_c == 4 && w.tiles_n == 4 && n_rows * (int64_t)K >= (1 << 20))) return 0;
  if ((uint64_t)n_src * (uint64_t)Cin * 4u >= 0xffffffffull || (uint64_t)n_rows * (uint64_t)Cout * 4u >= 0xffffffffull) return 0;
  return ((int64_t)K * Cin * Cout) % 4 == 0 && Cout % 4 == 0;
}

// bytes of partial slabs the equal-pieces stage 1 writes: one (Cin, Cout) slab per (piece, group it touches) <= pieces + 8 K - 1
extern "C" size_t sv_sparse_conv_wgrad_planned_bytes(int K, int Cin, int Cout) {
  return ((size_t)(sv_wgrad_plan_pieces(Cin, Cout) + WGE_EIGHTHS * K) * Cin * Cout * sizeof(float) + 255) / 256 * 256;
}

static int wgrad_planned_run(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K, int Cin, int Cout,
                             const void* plan, void* partial, void* stream, WgradOut out, WgradReduceJob* defer) {
  const InNorm in = take_input_norm();
  SV_CHECK_ARG(X && nbr && dY && dW && plan && partial, "sparse_conv_wgrad_planned: null pointer");
  SV_CHECK_ARG(sv_wgrad_planned_applies(n_src, n_rows, K, Cin, Cout), "sparse_conv_wgrad_planned: not for this layer (ask sv_wgrad_planned_applies first)");
  SV_CHECK_ARG((uintptr_t)dW % 16 == 0 && (uintptr_t)partial % 16 == 0, "sparse_conv_wgrad_planned: 16-byte alignment");
  const int pieces = sv_wgrad_plan_pieces(Cin, Cout);
  const WgradPlanPtrs p = wgrad_plan_ptrs(const_cast<void*>(plan), pieces, K);
  const WgradShape w = wgrad_shape(n_rows, K, Cin, Cout);
  hipStream_t st = sv_stream(stream);
  static const int prio = getenv("SEEVCN_WGRAD_PRIO") ? atoi(getenv("SEEVCN_WGRAD_PRIO")) : 1;
  WgradArgs a{X, nbr, dY, static_cast<float*>(partial), n_rows, K, Cin, Cout, 0, 0, prio, n_src};
  a.in_coef = in.coef, a.in_relu = in.relu;
  const WgradPlanView pl{p.cut, p.slab0, p.pre, pieces, (int)wgrad_units_per_group(n_rows)};
  if (w.tiles_c == 4) launch_wgrad_eq<4, 4>(a, pl, st);
  else if (w.tiles_c == 2 && w.tiles_n == 4) launch_wgrad_eq<2, 4>(a, pl, st);
  else if (w.tiles_c == 2) launch_wgrad_eq<2, 2>(a, pl, st);
  else if (w.tiles_n == 2) launch_wgrad_eq<1, 2>(a, pl, st);
  else launch_wgrad_eq<1, 1>(a, pl, st);
  const int64_t slab = (int64_t)K * Cin * Cout;
  if (defer) {
    defer->partial = a.partial, defer->dW = dW, defer->slab = slab, defer->nslabs = 1, defer->runs = p.runs, defer->out = out;
  } else {
    hipLaunchKernelGGL(k_wgrad_reduce4, dim3(sv_div_up(slab / 4, 64)), dim3(256), 0, st, a.partial, 1, slab / 4, dW, out, p.runs);
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// dW as sv_sparse_conv_wgrad_strided writes it (stride_k = 0: contiguous (K, Cin, Cout)), stage 1 on the equal pieces of `plan` (sv_wgrad_plan_build of
// THIS table with sv_wgrad_plan_pieces(Cin, Cout) pieces); partial: sv_sparse_conv_wgrad_planned_bytes
extern "C" int sv_sparse_conv_wgrad_planned(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K, int Cin, int Cout,
                                            int64_t stride_k, int64_t stride_cin, int64_t stride_cout, const void* plan, void* partial, void* stream) {
  SV_CHECK_ARG((stride_k == 0 && stride_cin == 0 && stride_cout == 0) || (stride_k > 0 && stride_cin > 0 && stride_cout > 0),
               "sparse_conv_wgrad_planned: strides all zero (contiguous) or all positive");
  const WgradOut out = stride_k ? WgradOut{stride_k, stride_cin, stride_cout, Cin, Cout, 0} : WgradOut{0, 0, 0, Cin, Cout, 1};
  return wgrad_planned_run(X, n_src, nbr, dY, dW, n_rows, K, Cin, Cout, plan, partial, stream, out, nullptr);
}

// stage 1 only; *job as sv_sparse_conv_wgrad_stage1 writes it, for sv_sparse_conv_wgrad_reduce_batch
extern "C" int sv_sparse_conv_wgrad_planned_stage1(const float* X, int64_t n_src, const int32_t* nbr, const float* dY, float* dW, int64_t n_rows, int K, int Cin,
                                                   int Cout, int64_t stride_k, int64_t stride_cin, int64_t stride_cout, const void* plan, void* partial, int64_t* job,
                                                   void* stream) {
  SV_CHECK_ARG(job && stride_k > 0 && stride_cin > 0 && stride_cout > 0, "sparse_conv_wgrad_planned_stage1: bad arguments");
  WgradReduceJob d{};
  int rc = wgrad_planned_run(X, n_src, nbr, dY, dW, n_rows, K, Cin, Cout, plan, partial, stream, WgradOut{stride_k, stride_cin, stride_cout, Cin, Cout, 0}, &d);
  job[0] = (int64_t)(uintptr_t)d.partial, job[1] = (int64_t)(uintptr_t)d.dW, job[2] = d.slab, job[3] = d.nslabs;
  job[4] = stride_k, job[5] = stride_cin, job[6] = stride_cout, job[7] = Cin, job[8] = Cout, job[9] = (int64_t)(uintptr_t)d.runs;
  return rc;
}
