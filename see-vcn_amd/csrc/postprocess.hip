// VCN post-processing on the GPU: the reference does all of this on the CPU after the network
// (see/surface_completion/models/VCN.py:89-93, models/vcn/utils/sampling.py:8-41,83-100, SEE_VCN.py:244-265).
//   k_surface_select   : np.unique(partial) -> k-NN into the coarse cloud -> list(set(indices)) -> tile to surface_pts
//   k_largest_cluster  : open3d cluster_dbscan(eps, min_points<=2) -> bincount argmax -> tile to total_pts
//   k_points_near_set  : compute_point_cloud_distance(...) < thresh (replace_with_completed_pts)
// One workgroup per object; everything lives in LDS / registers, HBM traffic is the clouds in and the surface out.
#include <stdlib.h>

#include "common.h"

#define PP_MAXN 1024     // points per object (resample_num and the network's coarse size)
#define PP_THREADS 1024
#define PP_WAVES (PP_THREADS / SV_WAVE)
#define PP_CPL (PP_MAXN / SV_WAVE)   // candidates per lane in the k-NN selection

__device__ __forceinline__ bool lex_less(float ax, float ay, float az, float bx, float by, float bz) {
  // row order of np.unique(axis=0) on an (N,3) float array: field-wise float compare, -0.0 == 0.0
  if (ax != bx) return ax < bx;
  if (ay != by) return ay < by;
  return az < bz;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    unsigned lo = __shfl_xor((unsigned)v, off), hi = __shfl_xor((unsigned)(v >> 32), off);
    unsigned long long o = ((unsigned long long)hi << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}

// CPython 3.10 set insertion order for small non-negative ints (hash(v) == v): Objects/setobject.c set_add_entry /
// set_insert_clean / set_table_resize (LINEAR_PROBES 9, PERTURB_SHIFT 5, grow x4 when fill*5 >= mask*3).  The reference turns
// the k-NN index list into `list(set(surface_idx))` (sampling.py:37), so the order of the selected surface points -- and with
// it which points the tile-to-1024 repeats once more -- is this table order.  Run by one thread; tables are LDS int16.
__device__ void set_insert_clean(short* table, int mask, int h) {
  unsigned perturb = h;
  int i = h & mask;
  while (true) {
    if (table[i] < 0) { table[i] = (short)h; return; }
    if (i + 9 <= mask) {
      for (int j = 1; j <= 9; ++j)
        if (table[i + j] < 0) { table[i + j] = (short)h; return; }
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

struct SurfaceArgs {
  const float* partial;   // (B, n, 3)
  const float* complete;  // (B, m, 3)
  float* surface;         // (B, surface_pts, 3)
  int* n_selected;        // (B)
  int n, m, k, surface_pts;
  // scratch
  unsigned short* query;  // (B, PP_MAXN) partial index of the u-th unique row, lexicographic order
  int* nq;                // (B)
  int* first;             // (B, PP_MAXN) first position of each coarse index in the reference's extend() list
};

#define PP_INF 0x7fffffff

// Bitonic sort of one element per thread over a 1024-thread workgroup: partners closer than a wave exchange by shuffle, the
// 10 long-stride stages go through LDS.  The comparison must be a strict weak order; ties may land in any order.
template <class FS, class FL>
__device__ __forceinline__ void pp_bitonic_stages(int tid, FS exchange_shfl, FL exchange_lds) {
  for (int size = 2; size <= PP_THREADS; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const bool take_min = ((tid & size) == 0) == ((tid & stride) == 0);
      if (stride < SV_WAVE) exchange_shfl(stride, take_min);
      else exchange_lds(stride, take_min);
    }
  }
}

// ---- A: np.unique(partial_pc, axis=0): sort the rows lexicographically (field-wise float compare, -0.0 == 0.0), flag the
// first row of every run of equal rows, prefix-sum the flags -> query[u] = index of a copy of the u-th unique row.
// this object's cloud P (n rows) -> query[u] = index of a copy of the u-th distinct row (lexicographic order), *nq = their number;
// `first` (PP_MAXN ints, may be null) is reset to PP_INF for the k-NN pass
__device__ __forceinline__ void surface_prep_body(const float* P, int n, unsigned short* query, int* nq, int* first) {
  __shared__ float s_x[PP_THREADS], s_y[PP_THREADS], s_z[PP_THREADS];
  __shared__ int s_v[PP_THREADS];
  __shared__ int s_wcnt[PP_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float x = __builtin_inff(), y = 0.f, z = 0.f;          // padding rows sort to the end
  int v = -1;
  if (tid < n) x = P[tid * 3], y = P[tid * 3 + 1], z = P[tid * 3 + 2], v = tid;
  if (first) first[tid] = PP_INF;
#define PREP_KEEP(ox, oy, oz, ov)                                                                     \
  {                                                                                                   \
    const bool sw = take_min ? lex_less(ox, oy, oz, x, y, z) : lex_less(x, y, z, ox, oy, oz);         \
    x = sw ? ox : x, y = sw ? oy : y, z = sw ? oz : z, v = sw ? ov : v;                               \
  }
  pp_bitonic_stages(
      tid,
      [&](int stride, bool take_min) {
        const float ox = __shfl_xor(x, stride), oy = __shfl_xor(y, stride), oz = __shfl_xor(z, stride);
        const int ov = __shfl_xor(v, stride);
        PREP_KEEP(ox, oy, oz, ov)
      },
      [&](int stride, bool take_min) {
        __syncthreads();
        s_x[tid] = x, s_y[tid] = y, s_z[tid] = z, s_v[tid] = v;
        __syncthreads();
        const int o = tid ^ stride;
        const float ox = s_x[o], oy = s_y[o], oz = s_z[o];
        const int ov = s_v[o];
        PREP_KEEP(ox, oy, oz, ov)
      });
#undef PREP_KEEP
  __syncthreads();
  s_x[tid] = x, s_y[tid] = y, s_z[tid] = z;
  __syncthreads();
  const bool head = v >= 0 && (tid == 0 || !(s_x[tid - 1] == x && s_y[tid - 1] == y && s_z[tid - 1] == z));
  const unsigned long long vote = __ballot(head);
  if (lane == 0) s_wcnt[wave] = __popcll(vote);
  __syncthreads();
  int rank = __popcll(vote & ((1ull << lane) - 1)), total = 0;
  for (int w = 0; w < PP_WAVES; ++w) {
    rank += w < wave ? s_wcnt[w] : 0;
    total += s_wcnt[w];
  }
  if (head) query[rank] = (unsigned short)v;
  if (tid == 0) *nq = total;
}

__global__ __launch_bounds__(PP_THREADS) void k_surface_prep(SurfaceArgs a) {
  const int b = blockIdx.x;
  surface_prep_body(a.partial + (size_t)b * a.n * 3, a.n, a.query + (size_t)b * PP_MAXN, a.nq + b, a.first + (size_t)b * PP_MAXN);
}

// wave-wide unsigned min on the DPP network (quad_perm xor1, xor2, row_ror 4/8, row_bcast 15/31), result from lane 63
__device__ __forceinline__ unsigned wave_umin_dpp(unsigned v) {
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xb1, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4e, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- B: k nearest coarse points per query in ascending distance (cKDTree.query(p, k)[1], float64 squared distances).
// One wave per query; lane owns coarse points lane, lane+64, ... : it sorts its 16 keys once (bitonic network in registers,
// sorted list parked in LDS), then k rounds of a wave-wide min over the lanes' heads pick the neighbours in order.
#define KNN_THREADS 256
#define KNN_WAVES (KNN_THREADS / SV_WAVE)
__global__ __launch_bounds__(KNN_THREADS) void k_surface_knn(SurfaceArgs a, int nsplit) {
  __shared__ float s_c[PP_MAXN * 3];
  __shared__ unsigned long long s_sorted[KNN_WAVES][PP_CPL][SV_WAVE];
  __shared__ int s_first[PP_MAXN];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* C = a.complete + (size_t)b * a.m * 3;
  const float* P = a.partial + (size_t)b * a.n * 3;
  for (int i = tid; i < a.m * 3; i += KNN_THREADS) s_c[i] = C[i];
  for (int i = tid; i < PP_MAXN; i += KNN_THREADS) s_first[i] = PP_INF;
  __syncthreads();
  const int nq = a.nq[b];
  unsigned long long(*mine)[SV_WAVE] = s_sorted[wave];
  for (int u = blockIdx.x * KNN_WAVES + wave; u < nq; u += nsplit * KNN_WAVES) {
    const int qi = a.query[(size_t)b * PP_MAXN + u];
    const double qx = (double)P[qi * 3], qy = (double)P[qi * 3 + 1], qz = (double)P[qi * 3 + 2];
    unsigned long long d[PP_CPL];
    unsigned tg[PP_CPL];
#pragma unroll
    for (int t = 0; t < PP_CPL; ++t) {
      const int c = lane + SV_WAVE * t;
      const double dx = (double)s_c[c * 3] - qx, dy = (double)s_c[c * 3 + 1] - qy, dz = (double)s_c[c * 3 + 2] - qz;
      const double dd = dx * dx + dy * dy + dz * dz;
      d[t] = (c < a.m) ? (unsigned long long)__double_as_longlong(dd) : ~0ull;   // dd >= 0: bit order == value order
      tg[t] = t;
    }
    // bitonic sort of (key, slot) ascending; equal keys keep the lower slot first (= lower coarse index)
#pragma unroll
    for (int size = 2; size <= PP_CPL; size <<= 1) {
#pragma unroll
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
        for (int i = 0; i < PP_CPL; ++i) {
          const int j = i ^ stride;
          if (j > i) {
            const bool up = (i & size) == 0;
            const bool lt = d[j] < d[i] || (d[j] == d[i] && tg[j] < tg[i]);
            const bool sw = up ? lt : !lt;
            const unsigned long long ki = d[i], kj = d[j];
            const unsigned ti = tg[i], tj = tg[j];
            d[i] = sw ? kj : ki, d[j] = sw ? ki : kj;
            tg[i] = sw ? tj : ti, tg[j] = sw ? ti : tj;
          }
        }
      }
    }
    unsigned long long tags = 0;
#pragma unroll
    for (int t = 0; t < PP_CPL; ++t) {
      mine[t][lane] = d[t];
      tags |= (unsigned long long)tg[t] << (4 * t);
    }
    unsigned long long cur = d[0];
    int head = 0;
    for (int r = 0; r < a.k; ++r) {
      const unsigned hi = (unsigned)(cur >> 32), lo = (unsigned)cur;
      const unsigned mhi = wave_umin_dpp(hi);
      unsigned long long vote = __ballot(hi == mhi);
      int wl;
      if (__popcll(vote) == 1) {                       // almost always: the top 32 bits of the float64 already decide
        wl = __ffsll((long long)vote) - 1;
      } else {
        const unsigned mlo = wave_umin_dpp(hi == mhi ? lo : 0xffffffffu);
        const bool match = hi == mhi && lo == mlo;
        vote = __ballot(match);
        if (__popcll(vote) == 1) {
          wl = __ffsll((long long)vote) - 1;
        } else {                                       // exact distance tie between lanes: lowest coarse index
          wl = (int)(wave_umin_dpp(match ? (unsigned)((tags & 15) * SV_WAVE + lane) : 0xffffffffu) & (SV_WAVE - 1));
        }
      }
      if (lane == wl) {
        atomicMin(&s_first[(int)(tags & 15) * SV_WAVE + lane], u * a.k + r);
        ++head;
        tags >>= 4;
        cur = head < PP_CPL ? mine[head][lane] : ~0ull;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < a.m; i += KNN_THREADS)
    if (s_first[i] != PP_INF) atomicMin(&a.first[(size_t)b * PP_MAXN + i], s_first[i]);
}

// ---- C: distinct indices in first-occurrence order -> CPython's set order -> np.tile(sel, [surface_pts, 1])[:surface_pts]
// Table sizes follow the key count alone (8 -> 32 after 5 keys -> 128 after 19 -> 512 after 77 -> 2048 after 307): all four
// small tables are cleared up front by the whole workgroup, one thread replays inserts and rehashes, and the final table is
// compacted by everyone.  >= 307 keys: every key < 1024 sits in its own slot of the 2048 table -> ascending order, no replay.
__device__ void set_replay(const unsigned short* seq, int n, short* t8, short* t32, short* t128, short* t512) {
  short* tabs[4] = {t8, t32, t128, t512};
  const int masks[4] = {7, 31, 127, 511};
  int level = 0, fill = 0;
  short* table = tabs[0];
  int mask = masks[0];
  for (int s = 0; s < n; ++s) {
    set_insert_clean(table, mask, seq[s]);           // keys are distinct: add == insert into a free slot on the same probe path
    ++fill;
    if (fill * 5 >= mask * 3 && level < 3) {
      short* nt = tabs[level + 1];
      const int nmask = masks[level + 1];
      for (int q = 0; q <= mask; ++q)
        if (table[q] >= 0) set_insert_clean(nt, nmask, table[q]);
      table = nt, mask = nmask, ++level;
    }
  }
}

__global__ __launch_bounds__(PP_THREADS) void k_surface_finish(SurfaceArgs a) {
  __shared__ float s_c[PP_MAXN * 3];
  __shared__ int s_first[PP_MAXN];
  __shared__ unsigned short s_seq[PP_MAXN];
  __shared__ unsigned short s_order[PP_MAXN];
  __shared__ short s_tab[8 + 32 + 128 + 512];
  __shared__ int s_nsel, s_wcnt[PP_WAVES];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* C = a.complete + (size_t)b * a.m * 3;
  for (int i = tid; i < a.m * 3; i += PP_THREADS) s_c[i] = C[i];
  s_first[tid] = tid < a.m ? a.first[(size_t)b * PP_MAXN + tid] : PP_INF;
  if (tid < 8 + 32 + 128 + 512) s_tab[tid] = -1;
  if (tid == 0) s_nsel = 0;
  __syncthreads();
  const int mine = s_first[tid];
  // selected indices in first-occurrence order: sort (first position, index); unselected (PP_INF) sink to the end
  int key = mine, val = tid;
#define FIN_KEEP(ok, ov)                                   \
  {                                                        \
    const bool sw = take_min ? ok < key : key < ok;        \
    key = sw ? ok : key, val = sw ? ov : val;              \
  }
  pp_bitonic_stages(
      tid,
      [&](int stride, bool take_min) {
        const int ok = __shfl_xor(key, stride), ov = __shfl_xor(val, stride);
        FIN_KEEP(ok, ov)
      },
      [&](int stride, bool take_min) {
        __syncthreads();
        s_first[tid] = key, s_seq[tid] = (unsigned short)val;
        __syncthreads();
        const int ok = s_first[tid ^ stride], ov = s_seq[tid ^ stride];
        FIN_KEEP(ok, ov)
      });
#undef FIN_KEEP
  __syncthreads();
  s_seq[tid] = (unsigned short)val;
  const unsigned long long sel_vote = __ballot(mine != PP_INF);
  if (lane == 0) s_wcnt[wave] = __popcll(sel_vote);
  __syncthreads();
  int rank_value = __popcll(sel_vote & ((1ull << lane) - 1)), nsel_total = 0;
  for (int w = 0; w < PP_WAVES; ++w) {
    rank_value += w < wave ? s_wcnt[w] : 0;
    nsel_total += s_wcnt[w];
  }
  if (tid == 0) s_nsel = nsel_total;
  __syncthreads();
  const int nsel = s_nsel;
  if (nsel >= 307) {
    if (mine != PP_INF) s_order[rank_value] = (unsigned short)tid;
  } else {
    if (tid == 0) set_replay(s_seq, nsel, s_tab, s_tab + 8, s_tab + 40, s_tab + 168);
    __syncthreads();
    const int off = nsel < 5 ? 0 : (nsel < 19 ? 8 : (nsel < 77 ? 40 : 168));       // final table by key count
    const int size = nsel < 5 ? 8 : (nsel < 19 ? 32 : (nsel < 77 ? 128 : 512));
    const short e = tid < size ? s_tab[off + tid] : (short)-1;
    const unsigned long long vote = __ballot(e >= 0);
    if (lane == 0) s_wcnt[wave] = __popcll(vote);
    __syncthreads();
    if (e >= 0) {
      int pos = __popcll(vote & ((1ull << lane) - 1));
      for (int w = 0; w < wave; ++w) pos += s_wcnt[w];
      s_order[pos] = (unsigned short)e;
    }
  }
  if (tid == 0) a.n_selected[b] = nsel;
  __syncthreads();
  float* out = a.surface + (size_t)b * a.surface_pts * 3;
  if (nsel > 0) {
    for (int i = tid; i < a.surface_pts * 3; i += PP_THREADS) {
      const int row = i / 3, col = i - row * 3;
      out[i] = s_c[s_order[row % nsel] * 3 + col];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// get_largest_cluster (sampling.py:83-100): open3d ClusterDBSCAN with min_points <= 2 has no border points, so clusters are
// the connected components of the strict (d^2 < eps^2, float64) neighbourhood graph among points that have >= min_points
// neighbours counting themselves; labels number the components by their first point, np.argmax(np.bincount) takes the
// first largest.  Union-find in LDS with the smaller index as root gives exactly that numbering.
struct ClusterArgs {
  const float* points;  // (B, n, 3)
  float* out;           // (B, total, 3)
  int* n_cluster;       // (B)
  int n, total, min_points;
  double eps2;
  const int* period;    // optional (B): the cloud of object b repeats its first period[b] rows cyclically (np.tile(...)[:n], what the surface selection
                        // returns); a HINT: verified on the device, a cloud that does not repeat takes the all-pairs path
};

__device__ __forceinline__ int uf_find(volatile int* parent, int x) {
  int p = parent[x];
  while (p != x) {
    const int g = parent[p];
    if (g != p) parent[x] = g;       // path halving (parents only ever decrease)
    x = p, p = g;
  }
  return x;
}

__global__ __launch_bounds__(PP_THREADS) void k_largest_cluster(ClusterArgs a) {
  __shared__ float s_x[PP_MAXN * 3];
  __shared__ int s_parent[PP_MAXN];
  __shared__ int s_cnt[PP_MAXN];
  __shared__ unsigned char s_core[PP_MAXN];
  __shared__ unsigned short s_member[PP_MAXN];
  __shared__ unsigned long long s_best;
  __shared__ int s_nmember;

  const int b = blockIdx.x, tid = threadIdx.x;
  const float* X = a.points + (size_t)b * a.n * 3;
  for (int i = tid; i < a.n * 3; i += PP_THREADS) s_x[i] = X[i];
  if (tid < PP_MAXN) s_parent[tid] = tid, s_cnt[tid] = 0, s_core[tid] = 0;
  if (tid == 0) s_best = 0ull, s_nmember = 0;
  __syncthreads();

  // Periodic clouds.  The surface selection tiles its U selected points to n rows (sampling.py:37-39: np.tile(sel, ...)[:surface_pts], U ~ 150-400 of
  // 1024), so row i equals row i - U: a copy is at distance 0 < eps from its original -- same component, both core -- and every other test between
  // copies repeats a test between originals.  The pair tests then run over the first U rows only (U^2 / 2 instead of n^2 / 2 float64 distances: 117 ->
  // ~10 us for 64 objects), the copies hang themselves under their originals; roots (smallest index of a component), core flags and member counts come
  // out as with all pairs.
  int U = a.n;
  if (a.period) {
    const int u = a.period[b];
    if (u >= 1 && u < a.n) {
      bool same = true;
      if (tid >= u && tid < a.n)
        same = __float_as_uint(s_x[tid * 3]) == __float_as_uint(s_x[(tid - u) * 3]) && __float_as_uint(s_x[tid * 3 + 1]) == __float_as_uint(s_x[(tid - u) * 3 + 1]) &&
               __float_as_uint(s_x[tid * 3 + 2]) == __float_as_uint(s_x[(tid - u) * 3 + 2]);
      if (__syncthreads_and(same ? 1 : 0)) U = u;
    }
  }
  if (tid >= U && tid < a.n) {                // a copy: joins its original (the smaller index stays the root), both are core
    s_parent[tid] = tid % U;
    s_core[tid] = 1, s_core[tid % U] = 1;
  }
  __syncthreads();

  if (tid < U) {
    const double x = s_x[tid * 3], y = s_x[tid * 3 + 1], z = s_x[tid * 3 + 2];
    bool has_nbr = false;
    const int half = U / 2;
    int my_parent = tid;
#pragma unroll 4
    for (int s = 1; s <= half; ++s) {         // every unordered pair once (twice for s == U/2 when U is even: harmless)
      int j = tid + s;
      if (j >= U) j -= U;
      // parent[j] is fetched with the coordinates so the common "already merged" test needs no dependent LDS round trip;
      // a stale value only sends us down the full union path, which re-reads
      const float jx = s_x[j * 3], jy = s_x[j * 3 + 1], jz = s_x[j * 3 + 2];
      const int pj = s_parent[j];
      const double dx = (double)jx - x, dy = (double)jy - y, dz = (double)jz - z;
      if (dx * dx + dy * dy + dz * dz < a.eps2) {
        has_nbr = true;
        s_core[j] = 1;
        if (pj != my_parent) {
          int ra = tid, rb = j;
          while (true) {
            ra = uf_find(s_parent, ra), rb = uf_find(s_parent, rb);
            if (ra == rb) break;
            if (ra > rb) { const int t = ra; ra = rb; rb = t; }
            if (atomicCAS(&s_parent[rb], rb, ra) == rb) break;
          }
          my_parent = uf_find(s_parent, tid);
        }
      }
    }
    if (has_nbr) s_core[tid] = 1;
  }
  __syncthreads();
  int root = -1;
  if (tid < a.n && (s_core[tid] || a.min_points <= 1)) {
    root = uf_find(s_parent, tid);
    atomicAdd(&s_cnt[root], 1);
  }
  __syncthreads();
  if (tid < a.n && s_cnt[tid] > 0)            // largest count, ties -> smallest root (first label)
    atomicMax(&s_best, ((unsigned long long)s_cnt[tid] << 32) | (unsigned)(PP_MAXN - tid));
  __syncthreads();
  const unsigned long long best = s_best;
  if (best == 0ull) {                          // every point is noise: the reference's np.argmax(np.bincount([])) raises
    if (tid == 0) a.n_cluster[b] = 0;
    return;
  }
  const int best_root = PP_MAXN - (int)(best & 0xffffffffu), count = (int)(best >> 32);
  // members in ascending index order (np.argwhere(labels == value))
  const bool mine = root == best_root;
  const unsigned long long vote = __ballot(mine);
  __shared__ int s_wave_cnt[PP_WAVES];
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 0) s_wave_cnt[wave] = __popcll(vote);
  __syncthreads();
  if (mine) {
    int pos = __popcll(vote & ((1ull << lane) - 1));
    for (int w = 0; w < wave; ++w) pos += s_wave_cnt[w];
    s_member[pos] = (unsigned short)tid;
  }
  __syncthreads();
  float* out = a.out + (size_t)b * a.total * 3;
  for (int i = tid; i < a.total * 3; i += PP_THREADS) {
    const int row = i / 3, col = i - row * 3;
    out[i] = s_x[s_member[row % count] * 3 + col];
  }
  if (tid == 0) a.n_cluster[b] = count;
}

// ---------------------------------------------------------------------------------------------------------------------
// replace_with_completed_pts (SEE_VCN.py:247-265): near[i] = (min_j |q_i - r_j| < thresh), float64 like open3d's
// compute_point_cloud_distance (sqrt of the squared NN distance compared with thresh).  The completed points arrive
// row-sorted (np.unique), so a tile of 256 consecutive rows spans a thin x-slab: a wave skips tiles whose slab is farther
// than thresh from all of its queries.
#define NS_TILE 256
// D = 3: rows [x,y,z]; D = 4: rows [b,x,y,z] and only points of the same scene b are compared (a batch of scenes in one launch;
// the merged instances are row-sorted, so tiles are scene-pure except at the seams).
template <int D>
__global__ __launch_bounds__(256) void k_points_near_set(const float* __restrict__ q, long nq, const float* __restrict__ r, long nr,
                                                       double thresh, unsigned char* __restrict__ near) {
  __shared__ float s_r[NS_TILE * D];
  __shared__ float s_wlo[4][D], s_whi[4][D];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < nq;
  float f[D];
#pragma unroll
  for (int c = 0; c < D; ++c) f[c] = live ? q[i * D + c] : 0.f;
  const double x = f[D - 3], y = f[D - 2], z = f[D - 1];
  bool found = !live;
  for (long base = 0; base < nr; base += NS_TILE) {
    const int cnt = (int)((nr - base) < NS_TILE ? (nr - base) : NS_TILE);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * D; t += 256) s_r[t] = r[base * D + t];
    __syncthreads();
    {                                            // tile bounding box: wave shuffles, then 4 partials
      const int t = threadIdx.x < cnt ? threadIdx.x : 0;
      const bool dropped = D == 4 && s_r[t * D] < 0.f;          // rows with scene id -1 (sv_dedup_rows) are nobody's neighbour: not in the box
#pragma unroll
      for (int c = 0; c < D; ++c) {
        float lo = dropped ? INFINITY : s_r[t * D + c], hi = dropped ? -INFINITY : s_r[t * D + c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          lo = fminf(lo, __shfl_xor(lo, off));
          hi = fmaxf(hi, __shfl_xor(hi, off));
        }
        if ((threadIdx.x & 63) == 0) s_wlo[threadIdx.x >> 6][c] = lo, s_whi[threadIdx.x >> 6][c] = hi;
      }
    }
    __syncthreads();
    float blo[D], bhi[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
      blo[c] = fminf(fminf(s_wlo[0][c], s_wlo[1][c]), fminf(s_wlo[2][c], s_wlo[3][c]));
      bhi[c] = fmaxf(fmaxf(s_whi[0][c], s_whi[1][c]), fmaxf(s_whi[2][c], s_whi[3][c]));
    }
    const double ex = x < (double)blo[D - 3] ? (double)blo[D - 3] - x : (x > (double)bhi[D - 3] ? x - (double)bhi[D - 3] : 0.0);
    const double ey = y < (double)blo[D - 2] ? (double)blo[D - 2] - y : (y > (double)bhi[D - 2] ? y - (double)bhi[D - 2] : 0.0);
    const double ez = z < (double)blo[D - 1] ? (double)blo[D - 1] - z : (z > (double)bhi[D - 1] ? z - (double)bhi[D - 1] : 0.0);
    bool maybe = !found && sqrt(ex * ex + ey * ey + ez * ez) < thresh;   // box distance <= point distance
    if (D == 4) maybe = maybe && f[0] >= blo[0] && f[0] <= bhi[0];
    if (__ballot(maybe) == 0ull) continue;
    if (maybe) {
      for (int t = 0; t < cnt; ++t) {
        if (D == 4 && s_r[t * D] != f[0]) continue;
        const double dx = x - (double)s_r[t * D + D - 3], dy = y - (double)s_r[t * D + D - 2], dz = z - (double)s_r[t * D + D - 1];
        if (sqrt(dx * dx + dy * dy + dz * dz) < thresh) { found = true; break; }
      }
    }
  }
  if (live) near[i] = found ? 1 : 0;
}

// Exact-duplicate detection per object (ResamplePoints tiles Ni points to 1024: every per-point layer of VCN only needs the
// distinct rows).  Same sort as k_surface_prep; writes int32 indices and counts.
__global__ __launch_bounds__(PP_THREADS) void k_unique_rows(const float* __restrict__ x, int n, int32_t* __restrict__ uniq_idx,
                                                          int32_t* __restrict__ counts) {
  __shared__ unsigned short s_q[PP_MAXN];
  __shared__ int s_n;
  surface_prep_body(x + (size_t)blockIdx.x * n * 3, n, s_q, &s_n, nullptr);
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid < s_n) uniq_idx[(size_t)blockIdx.x * n + tid] = s_q[tid];
  if (tid == 0) counts[blockIdx.x] = s_n;
}

extern "C" int sv_unique_rows(const float* x, int batch, int n, int32_t* uniq_idx, int32_t* counts, void* stream) {
  SV_CHECK_ARG(batch >= 0 && n >= 1 && n <= PP_MAXN, "sv_unique_rows: 1 <= n <= %d (got %d)", PP_MAXN, n);
  if (batch == 0) return SV_OK;
  SV_CHECK_ARG(x && uniq_idx && counts, "sv_unique_rows: null pointer");
  hipLaunchKernelGGL(k_unique_rows, dim3(batch), dim3(PP_THREADS), 0, sv_stream(stream), x, n, uniq_idx, counts);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// Packs the per-object lists of sv_unique_rows behind each other: sel[off_b + r] = b*n + uniq_idx[b][r], row_group[off_b + r] = b
// with off_b = counts[0] + .. + counts[b-1]; *total = all kept rows.  One workgroup per object (it sums the counts before it).
__global__ __launch_bounds__(256) void k_unique_compact(const int32_t* __restrict__ uniq_idx, const int32_t* __restrict__ counts, int batch,
                                                        int n, int64_t* __restrict__ sel, int32_t* __restrict__ row_group,
                                                        int32_t* __restrict__ total) {
  __shared__ int s_part[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int acc = 0;
  for (int i = tid; i < b; i += 256) acc += counts[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) s_part[wave] = acc;
  __syncthreads();
  const int first = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  const int cnt = counts[b];
  for (int r = tid; r < cnt; r += 256) {
    sel[first + r] = (int64_t)b * n + uniq_idx[(size_t)b * n + r];
    row_group[first + r] = b;
  }
  if (b == batch - 1 && tid == 0) *total = first + cnt;
}

extern "C" int sv_unique_rows_compact(const int32_t* uniq_idx, const int32_t* counts, int batch, int n, int64_t* sel, int32_t* row_group,
                                      int32_t* total, void* stream) {
  SV_CHECK_ARG(batch >= 0 && n >= 1, "sv_unique_rows_compact: bad sizes");
  if (batch == 0) return SV_OK;
  SV_CHECK_ARG(uniq_idx && counts && sel && row_group && total, "sv_unique_rows_compact: null pointer");
  hipLaunchKernelGGL(k_unique_compact, dim3(batch), dim3(256), 0, sv_stream(stream), uniq_idx, counts, batch, n, sel, row_group, total);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" size_t sv_vcn_surface_select_scratch_bytes(int batch) {
  return (size_t)(batch < 1 ? 1 : batch) * (PP_MAXN * (sizeof(int) + sizeof(unsigned short)) + 2 * sizeof(int));
}

extern "C" int sv_vcn_surface_select(const float* partial, const float* complete, int batch, int n_partial, int n_complete, int k,
                                     int surface_pts, void* scratch, float* surface, int32_t* n_selected, void* stream) {
  SV_CHECK_ARG(batch >= 0 && n_partial >= 1 && n_partial <= PP_MAXN && n_complete >= 1 && n_complete <= PP_MAXN,
               "sv_vcn_surface_select: 1 <= n_partial, n_complete <= %d (got %d, %d)", PP_MAXN, n_partial, n_complete);
  SV_CHECK_ARG(k >= 1 && k <= n_complete, "sv_vcn_surface_select: 1 <= k <= n_complete (k=%d)", k);
  SV_CHECK_ARG(surface_pts >= 1, "sv_vcn_surface_select: surface_pts >= 1");
  if (batch == 0) return SV_OK;
  SV_CHECK_ARG(partial && complete && surface && n_selected && scratch, "sv_vcn_surface_select: null pointer");
  SurfaceArgs a{partial, complete, surface, n_selected, n_partial, n_complete, k, surface_pts, nullptr, nullptr, nullptr};
  a.first = reinterpret_cast<int*>(scratch);
  a.nq = a.first + (size_t)batch * PP_MAXN;
  a.query = reinterpret_cast<unsigned short*>(a.nq + batch + (batch & 1));
  // enough query slices per object to give every CU a workgroup
  static const int nsplit_env = getenv("SEEVCN_KNN_SPLIT") ? atoi(getenv("SEEVCN_KNN_SPLIT")) : 0;     // measurement switch
  // (64 objects: 16 slices 0.295 ms for the three launches, 32 slices 0.228, 64 slices 0.208 -- a slice's fixed cost, the 12 KB coarse cloud into LDS, is small
  // beside a query's 16 float64 distances per lane, 80 compare-exchanges and k wave-min rounds)
  int nsplit = 4096 / batch;
  nsplit = nsplit < 1 ? 1 : (nsplit > 64 ? 64 : nsplit);
  if (nsplit_env > 0) nsplit = nsplit_env;
  hipLaunchKernelGGL(k_surface_prep, dim3(batch), dim3(PP_THREADS), 0, sv_stream(stream), a);
  hipLaunchKernelGGL(k_surface_knn, dim3(nsplit, batch), dim3(KNN_THREADS), 0, sv_stream(stream), a, nsplit);
  hipLaunchKernelGGL(k_surface_finish, dim3(batch), dim3(PP_THREADS), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

static int largest_cluster_run(const float* points, int batch, int n, const int32_t* period, double eps, int min_points, int total_pts, float* out,
                               int32_t* n_cluster, void* stream);
extern "C" int sv_vcn_largest_cluster(const float* points, int batch, int n, double eps, int min_points, int total_pts, float* out,
                                      int32_t* n_cluster, void* stream) {
  return largest_cluster_run(points, batch, n, nullptr, eps, min_points, total_pts, out, n_cluster, stream);
}
// the same with a hint: object b's cloud repeats its first period[b] rows (device int32, the surface selection's n_selected); see k_largest_cluster
extern "C" int sv_vcn_largest_cluster_periodic(const float* points, int batch, int n, const int32_t* period, double eps, int min_points, int total_pts,
                                               float* out, int32_t* n_cluster, void* stream) {
  return largest_cluster_run(points, batch, n, period, eps, min_points, total_pts, out, n_cluster, stream);
}
static int largest_cluster_run(const float* points, int batch, int n, const int32_t* period, double eps, int min_points, int total_pts, float* out,
                               int32_t* n_cluster, void* stream) {
  SV_CHECK_ARG(batch >= 0 && n >= 1 && n <= PP_MAXN, "sv_vcn_largest_cluster: 1 <= n <= %d (got %d)", PP_MAXN, n);
  SV_CHECK_ARG(min_points >= 0 && min_points <= 2,
               "sv_vcn_largest_cluster: min_points <= 2 only (border-point order of open3d's BFS is not reproduced); got %d", min_points);
  SV_CHECK_ARG(total_pts >= 1 && eps > 0, "sv_vcn_largest_cluster: total_pts >= 1, eps > 0");
  if (batch == 0) return SV_OK;
  SV_CHECK_ARG(points && out && n_cluster, "sv_vcn_largest_cluster: null pointer");
  ClusterArgs a{points, out, n_cluster, n, total_pts, min_points, eps * eps, period};
  hipLaunchKernelGGL(k_largest_cluster, dim3(batch), dim3(PP_THREADS), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Exact duplicates among rows [b,x,y,z] (np.unique(np.vstack(instances), axis=0) of SEE_VCN.py:115, without the sort): every row goes
// into an open-addressing table of row indices; rows that meet an equal row there settle on the SMALLEST index among them (atomicMin on
// the slot: all candidates are equal rows, so a probe that compares against any of them decides the same).  A second pass gives every
// row that is not its slot's final owner b = -1, which every consumer downstream ignores (k_points_near_set: scene ids must match and
// dropped rows stay out of the tile boxes; the voxelisers: scene id out of range).  Keeping the FIRST copy matters for speed, not for the
// result: the cluster output repeats its points cyclically, so the survivors are the leading rows of each object and the tiles behind them
// are dropped whole.  Float compare like numpy's: -0.0 == 0.0, NaN equals nothing.
__global__ __launch_bounds__(256) void k_dedup_insert(const float* __restrict__ rows, int64_t n, int32_t* __restrict__ table, unsigned int tmask,
                                                      int32_t* __restrict__ slot_of) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 v = reinterpret_cast<const float4*>(rows)[i];
  if (v.x < 0.f) { slot_of[i] = -1; return; }                  // already dropped
  unsigned long long h = 0x9E3779B97F4A7C15ull;
  const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const unsigned bits = f[c] == 0.f ? 0u : __float_as_uint(f[c]);
    h = (h ^ bits) * 0xFF51AFD7ED558CCDull;
    h ^= h >> 29;
  }
  unsigned int slot = (unsigned int)(h >> 20) & tmask;
  while (true) {
    const int32_t prev = atomicCAS(&table[slot], -1, (int32_t)i);
    if (prev == -1) break;
    const float4 o = reinterpret_cast<const float4*>(rows)[prev];
    if (o.x == v.x && o.y == v.y && o.z == v.z && o.w == v.w) {
      atomicMin(&table[slot], (int32_t)i);
      break;
    }
    slot = (slot + 1) & tmask;
  }
  slot_of[i] = (int32_t)slot;
}

__global__ __launch_bounds__(256) void k_dedup_flag(float* __restrict__ rows, int64_t n, const int32_t* __restrict__ table,
                                                    const int32_t* __restrict__ slot_of) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t s = slot_of[i];
  if (s >= 0 && table[s] != (int32_t)i) rows[i * 4] = -1.f;
}

extern "C" size_t sv_dedup_rows_scratch_bytes(int64_t n) {
  int64_t t = 1024;
  while (t < 2 * n) t <<= 1;
  return (size_t)t * 4 + (size_t)(n > 0 ? n : 1) * 4;
}

extern "C" int sv_dedup_rows(float* rows, int64_t n, void* scratch, void* stream) {
  SV_CHECK_ARG(n >= 0 && n < (int64_t)1 << 30, "sv_dedup_rows: bad size");
  if (n == 0) return SV_OK;
  SV_CHECK_ARG(rows && scratch && ((uintptr_t)rows % 16 == 0), "sv_dedup_rows: null or unaligned pointer");
  int64_t t = 1024;
  while (t < 2 * n) t <<= 1;
  hipStream_t st = sv_stream(stream);
  int32_t* table = static_cast<int32_t*>(scratch);
  int32_t* slot_of = table + t;
  SV_HIP(hipMemsetAsync(table, 0xFF, (size_t)t * 4, st));
  hipLaunchKernelGGL(k_dedup_insert, dim3(sv_div_up(n, 256)), dim3(256), 0, st, rows, n, table, (unsigned int)(t - 1), slot_of);
  hipLaunchKernelGGL(k_dedup_flag, dim3(sv_div_up(n, 256)), dim3(256), 0, st, rows, n, table, slot_of);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// The same test without a workgroup-wide tile loop: the boxes of the reference tiles (NB_TILE rows each, dropped rows left out) are made
// ONCE by k_near_boxes; a wave then walks the boxes -- 2 x D floats at a wave-uniform address -- and opens a tile only if one of its 64
// queries is within thresh of the box, reading the tile's rows at wave-uniform addresses as well (scalar loads: no LDS, no barrier).
// k_points_near_set above pays two barriers, an LDS fill and a box reduction per (workgroup, tile) whether or not the tile is opened:
// 1.26 ms for 334 k queries x 65 k unsorted reference rows, 77 us when the rows arrive sorted and unique (fewer, thinner tiles).
#define NB_TILE 128
template <int D>
__global__ __launch_bounds__(NB_TILE) void k_near_boxes(const float* __restrict__ r, long nr, float* __restrict__ boxes) {
  __shared__ float s_lo[2][D], s_hi[2][D];
  const long i = (long)blockIdx.x * NB_TILE + threadIdx.x;
  const bool live = i < nr && !(D == 4 && r[i * D] < 0.f);
#pragma unroll
  for (int c = 0; c < D; ++c) {
    float lo = live ? r[i * D + c] : INFINITY, hi = live ? r[i * D + c] : -INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, off));
      hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0) s_lo[threadIdx.x >> 6][c] = lo, s_hi[threadIdx.x >> 6][c] = hi;
  }
  __syncthreads();
  if (threadIdx.x < D) {
    boxes[(long)blockIdx.x * 2 * D + threadIdx.x] = fminf(s_lo[0][threadIdx.x], s_lo[1][threadIdx.x]);
    boxes[(long)blockIdx.x * 2 * D + D + threadIdx.x] = fmaxf(s_hi[0][threadIdx.x], s_hi[1][threadIdx.x]);
  }
}

template <int D>
__global__ __launch_bounds__(256) void k_points_near_boxed(const float* __restrict__ q, long nq, const float* __restrict__ r, long nr,
                                                         const float* __restrict__ boxes, double thresh, unsigned char* __restrict__ near) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < nq;
  float f[D];
#pragma unroll
  for (int c = 0; c < D; ++c) f[c] = live ? q[i * D + c] : 0.f;
  const double x = f[D - 3], y = f[D - 2], z = f[D - 1];
  bool found = !live;
  const long ntiles = (nr + NB_TILE - 1) / NB_TILE;
  for (long tile = 0; tile < ntiles; ++tile) {
    const float* bx = boxes + tile * 2 * D;                    // wave-uniform
    float blo[D], bhi[D];
#pragma unroll
    for (int c = 0; c < D; ++c) blo[c] = bx[c], bhi[c] = bx[D + c];
    if (blo[D - 3] > bhi[D - 3]) continue;                     // no live row in the tile (all dropped): wave-uniform
    {
      // cheap float screen, on the safe side of the exact test below (slack well above the float rounding of metre-scale coordinates)
      const float fx = fmaxf(fmaxf(blo[D - 3] - f[D - 3], f[D - 3] - bhi[D - 3]), 0.f);
      const float fy = fmaxf(fmaxf(blo[D - 2] - f[D - 2], f[D - 2] - bhi[D - 2]), 0.f);
      const float fz = fmaxf(fmaxf(blo[D - 1] - f[D - 1], f[D - 1] - bhi[D - 1]), 0.f);
      const float lim = (float)thresh * 1.001f + 1e-3f;
      bool screen = !found && fx * fx + fy * fy + fz * fz < lim * lim;
      if (D == 4) screen = screen && f[0] >= blo[0] && f[0] <= bhi[0];
      if (__ballot(screen) == 0ull) continue;
    }
    const double ex = x < (double)blo[D - 3] ? (double)blo[D - 3] - x : (x > (double)bhi[D - 3] ? x - (double)bhi[D - 3] : 0.0);
    const double ey = y < (double)blo[D - 2] ? (double)blo[D - 2] - y : (y > (double)bhi[D - 2] ? y - (double)bhi[D - 2] : 0.0);
    const double ez = z < (double)blo[D - 1] ? (double)blo[D - 1] - z : (z > (double)bhi[D - 1] ? z - (double)bhi[D - 1] : 0.0);
    bool maybe = !found && sqrt(ex * ex + ey * ey + ez * ez) < thresh;   // box distance <= point distance
    if (D == 4) maybe = maybe && f[0] >= blo[0] && f[0] <= bhi[0];
    if (__ballot(maybe) == 0ull) continue;
    const long base = tile * NB_TILE;
    const int cnt = (int)((nr - base) < NB_TILE ? (nr - base) : NB_TILE);
    for (int t = 0; t < cnt; ++t) {                           // wave-uniform loop and addresses; lanes that are done just wait
      const float* row = r + (base + t) * D;
      float rv[D];
#pragma unroll
      for (int c = 0; c < D; ++c) rv[c] = row[c];
      if (maybe && !found && !(D == 4 && rv[0] != f[0])) {
        const double dx = x - (double)rv[D - 3], dy = y - (double)rv[D - 2], dz = z - (double)rv[D - 1];
        if (sqrt(dx * dx + dy * dy + dz * dz) < thresh) found = true;
      }
      if (__ballot(maybe && !found) == 0ull) break;
    }
  }
  if (live) near[i] = found ? 1 : 0;
}

extern "C" size_t sv_points_near_set_scratch_bytes(int64_t n_ref) { return (size_t)((n_ref > 0 ? n_ref : 0) / NB_TILE + 1) * 8 * sizeof(float); }

extern "C" int sv_points_near_set_boxed(const float* query, int64_t n_query, const float* ref, int64_t n_ref, int row_dim, double thresh, void* scratch,
                                        uint8_t* near, void* stream) {
  SV_CHECK_ARG(n_query >= 0 && n_ref >= 0 && thresh >= 0, "sv_points_near_set: negative size");
  SV_CHECK_ARG(row_dim == 3 || row_dim == 4, "sv_points_near_set: row_dim must be 3 ([x,y,z]) or 4 ([b,x,y,z]), got %d", row_dim);
  if (n_query == 0) return SV_OK;
  SV_CHECK_ARG(query && near && (n_ref == 0 || (ref && scratch)), "sv_points_near_set: null pointer");
  hipStream_t st = sv_stream(stream);
  float* boxes = static_cast<float*>(scratch);
  const int ntiles = (int)((n_ref + NB_TILE - 1) / NB_TILE);
  if (row_dim == 3) {
    if (ntiles) hipLaunchKernelGGL(k_near_boxes<3>, dim3(ntiles), dim3(NB_TILE), 0, st, ref, (long)n_ref, boxes);
    hipLaunchKernelGGL(k_points_near_boxed<3>, dim3(sv_div_up(n_query, 256)), dim3(256), 0, st, query, (long)n_query, ref, (long)n_ref, boxes, thresh, near);
  } else {
    if (ntiles) hipLaunchKernelGGL(k_near_boxes<4>, dim3(ntiles), dim3(NB_TILE), 0, st, ref, (long)n_ref, boxes);
    hipLaunchKernelGGL(k_points_near_boxed<4>, dim3(sv_div_up(n_query, 256)), dim3(256), 0, st, query, (long)n_query, ref, (long)n_ref, boxes, thresh, near);
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_points_near_set(const float* query, int64_t n_query, const float* ref, int64_t n_ref, int row_dim, double thresh,
                                  uint8_t* near, void* stream) {
  SV_CHECK_ARG(n_query >= 0 && n_ref >= 0 && thresh >= 0, "sv_points_near_set: negative size");
  SV_CHECK_ARG(row_dim == 3 || row_dim == 4, "sv_points_near_set: row_dim must be 3 ([x,y,z]) or 4 ([b,x,y,z]), got %d", row_dim);
  if (n_query == 0) return SV_OK;
  SV_CHECK_ARG(query && near && (ref || n_ref == 0), "sv_points_near_set: null pointer");
  if (row_dim == 3)
    hipLaunchKernelGGL(k_points_near_set<3>, dim3(sv_div_up(n_query, 256)), dim3(256), 0, sv_stream(stream), query, (long)n_query, ref,
                       (long)n_ref, thresh, near);
  else
    hipLaunchKernelGGL(k_points_near_set<4>, dim3(sv_div_up(n_query, 256)), dim3(256), 0, sv_stream(stream), query, (long)n_query, ref,
                       (long)n_ref, thresh, near);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
