// VCN post-processing on the GPU: the reference does all of this on the CPU after the network
// (see/surface_completion/models/VCN.py:89-93, models/vcn/utils/sampling.py:8-41,83-100, SEE_VCN.py:244-265).
//   k_surface_select   : np.unique(partial) -> k-NN into the coarse cloud -> list(set(indices)) -> tile to surface_pts
//   k_largest_cluster  : open3d cluster_dbscan(eps, min_points<=2) -> bincount argmax -> tile to total_pts
//   k_points_near_set  : compute_point_cloud_distance(...) < thresh (replace_with_completed_pts)
// One workgroup per object; everything lives in LDS / registers, HBM traffic is the clouds in and the surface out.
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

__device__ int cpython_set_order(const unsigned short* seq, int n, short* tab_a, short* tab_b, unsigned short* order) {
  short* table = tab_a;
  short* other = tab_b;
  int mask = 7, fill = 0;
  for (int i = 0; i < 8; ++i) table[i] = -1;
  for (int s = 0; s < n; ++s) {           // seq holds distinct values: every step is an insertion
    const int h = seq[s];
    unsigned perturb = h;
    int i = h & mask;
    bool done = false;
    while (!done) {
      const int probes = (i + 9 <= mask) ? 9 : 0;
      for (int j = 0; j <= probes; ++j) {
        if (table[i + j] < 0) {
          table[i + j] = (short)h;
          ++fill;
          done = true;
          break;
        }
      }
      if (!done) {
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & mask;
      }
    }
    if (fill * 5 >= mask * 3) {
      int newsize = 8;
      while (newsize <= fill * 4) newsize <<= 1;
      for (int q = 0; q < newsize; ++q) other[q] = -1;
      for (int q = 0; q <= mask; ++q)
        if (table[q] >= 0) set_insert_clean(other, newsize - 1, table[q]);
      short* t = table; table = other; other = t;
      mask = newsize - 1;
    }
  }
  int m = 0;
  for (int q = 0; q <= mask; ++q)
    if (table[q] >= 0) order[m++] = (unsigned short)table[q];
  return m;
}

struct SurfaceArgs {
  const float* partial;   // (B, n, 3)
  const float* complete;  // (B, m, 3)
  float* surface;         // (B, surface_pts, 3)
  int* n_selected;        // (B)
  int n, m, k, surface_pts;
};

__global__ __launch_bounds__(PP_THREADS) void k_surface_select(SurfaceArgs a) {
  __shared__ float s_p[PP_MAXN * 3];            // partial cloud
  __shared__ float s_c[PP_MAXN * 3];            // coarse (complete) cloud
  __shared__ unsigned short s_rep[PP_MAXN];     // 1 = first copy of its coordinates
  __shared__ unsigned short s_query[PP_MAXN];   // representative index by lexicographic rank
  __shared__ int s_first[PP_MAXN];              // first position of each coarse index in the reference's extend() list
  __shared__ unsigned short s_seq[PP_MAXN];     // distinct coarse indices in first-occurrence order
  __shared__ unsigned short s_order[PP_MAXN];   // ... in CPython set iteration order
  __shared__ short s_tab[2][2048];
  __shared__ int s_nq, s_nsel;

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* P = a.partial + (size_t)b * a.n * 3;
  const float* C = a.complete + (size_t)b * a.m * 3;
  for (int i = tid; i < a.n * 3; i += PP_THREADS) s_p[i] = P[i];
  for (int i = tid; i < a.m * 3; i += PP_THREADS) s_c[i] = C[i];
  if (tid < PP_MAXN) s_first[tid] = 0x7fffffff;
  if (tid == 0) s_nq = 0, s_nsel = 0;
  __syncthreads();

  // ---- np.unique(partial_pc, axis=0): representatives (earliest copy) and their lexicographic rank
  float px = 0, py = 0, pz = 0;
  bool rep = false;
  if (tid < a.n) {
    px = s_p[tid * 3], py = s_p[tid * 3 + 1], pz = s_p[tid * 3 + 2];
    rep = true;
    for (int j = 0; j < tid; ++j)
      if (s_p[j * 3] == px && s_p[j * 3 + 1] == py && s_p[j * 3 + 2] == pz) { rep = false; break; }
    s_rep[tid] = rep;
  }
  __syncthreads();
  if (rep) {
    int rank = 0;
    for (int j = 0; j < a.n; ++j)
      if (s_rep[j] && lex_less(s_p[j * 3], s_p[j * 3 + 1], s_p[j * 3 + 2], px, py, pz)) ++rank;
    s_query[rank] = (unsigned short)tid;
    atomicAdd(&s_nq, 1);
  }
  __syncthreads();
  const int nq = s_nq;

  // ---- k nearest coarse points per query in ascending distance (cKDTree.query(p, k)[1], float64 squared distances);
  //      one wave per query, lane owns coarse points lane, lane+64, ...; only the first position of each index is kept.
  for (int u = wave; u < nq; u += PP_WAVES) {
    const int qi = s_query[u];
    const double qx = (double)s_p[qi * 3], qy = (double)s_p[qi * 3 + 1], qz = (double)s_p[qi * 3 + 2];
    unsigned long long d[PP_CPL];
#pragma unroll
    for (int t = 0; t < PP_CPL; ++t) {
      const int c = lane + SV_WAVE * t;
      const double dx = (double)s_c[c * 3] - qx, dy = (double)s_c[c * 3 + 1] - qy, dz = (double)s_c[c * 3 + 2] - qz;
      const double dd = dx * dx + dy * dy + dz * dz;
      d[t] = (c < a.m) ? (unsigned long long)__double_as_longlong(dd) : ~0ull;   // dd >= 0: bit order == value order
    }
    for (int r = 0; r < a.k; ++r) {
      unsigned long long best = d[0];
      int bt = 0;
#pragma unroll
      for (int t = 1; t < PP_CPL; ++t)
        if (d[t] < best) best = d[t], bt = t;
      const unsigned long long wmin = wave_min_u64(best);
      int cand = best == wmin ? bt * SV_WAVE + lane : 0x7fffffff;     // ties: lowest coarse index
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off));
      if ((cand & (SV_WAVE - 1)) == lane) {
        atomicMin(&s_first[cand], u * a.k + r);
        const int wt = cand / SV_WAVE;
#pragma unroll
        for (int t = 0; t < PP_CPL; ++t)
          if (t == wt) d[t] = ~0ull;
      }
    }
  }
  __syncthreads();

  // ---- distinct indices in first-occurrence order, then CPython's set order
  if (tid < a.m && s_first[tid] != 0x7fffffff) {
    const int mine = s_first[tid];
    int rank = 0;
    for (int j = 0; j < a.m; ++j) rank += s_first[j] < mine;
    s_seq[rank] = (unsigned short)tid;
    atomicAdd(&s_nsel, 1);
  }
  __syncthreads();
  if (tid == 0) {
    const int m = cpython_set_order(s_seq, s_nsel, s_tab[0], s_tab[1], s_order);
    a.n_selected[b] = m;
  }
  __syncthreads();
  const int nsel = s_nsel;
  float* out = a.surface + (size_t)b * a.surface_pts * 3;
  if (nsel > 0) {
    for (int i = tid; i < a.surface_pts * 3; i += PP_THREADS) {      // np.tile(sel, [surface_pts, 1])[:surface_pts]
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

  if (tid < a.n) {
    const double x = s_x[tid * 3], y = s_x[tid * 3 + 1], z = s_x[tid * 3 + 2];
    bool has_nbr = false;
    const int half = a.n / 2;
    for (int s = 1; s <= half; ++s) {         // every unordered pair once (twice for s == n/2 when n is even: harmless)
      int j = tid + s;
      if (j >= a.n) j -= a.n;
      const double dx = (double)s_x[j * 3] - x, dy = (double)s_x[j * 3 + 1] - y, dz = (double)s_x[j * 3 + 2] - z;
      if (dx * dx + dy * dy + dz * dz < a.eps2) {
        has_nbr = true;
        s_core[j] = 1;
        int ra = tid, rb = j;
        while (true) {
          ra = uf_find(s_parent, ra), rb = uf_find(s_parent, rb);
          if (ra == rb) break;
          if (ra > rb) { const int t = ra; ra = rb; rb = t; }
          if (atomicCAS(&s_parent[rb], rb, ra) == rb) break;
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
__global__ __launch_bounds__(256) void k_points_near_set(const float* __restrict__ q, long nq, const float* __restrict__ r, long nr,
                                                       double thresh, unsigned char* __restrict__ near) {
  __shared__ float s_r[NS_TILE * 3];
  __shared__ float s_wlo[4][3], s_whi[4][3];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < nq;
  const float fx = live ? q[i * 3] : 0.f, fy = live ? q[i * 3 + 1] : 0.f, fz = live ? q[i * 3 + 2] : 0.f;
  const double x = fx, y = fy, z = fz;
  bool found = !live;
  for (long base = 0; base < nr; base += NS_TILE) {
    const int cnt = (int)((nr - base) < NS_TILE ? (nr - base) : NS_TILE);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * 3; t += 256) s_r[t] = r[base * 3 + t];
    __syncthreads();
    {                                            // tile bounding box: wave shuffles, then 4 partials
      const int t = threadIdx.x < cnt ? threadIdx.x : 0;
      float lo[3], hi[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        lo[c] = hi[c] = s_r[t * 3 + c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          lo[c] = fminf(lo[c], __shfl_xor(lo[c], off));
          hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], off));
        }
        if ((threadIdx.x & 63) == 0) s_wlo[threadIdx.x >> 6][c] = lo[c], s_whi[threadIdx.x >> 6][c] = hi[c];
      }
    }
    __syncthreads();
    double blo[3], bhi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      blo[c] = (double)fminf(fminf(s_wlo[0][c], s_wlo[1][c]), fminf(s_wlo[2][c], s_wlo[3][c]));
      bhi[c] = (double)fmaxf(fmaxf(s_whi[0][c], s_whi[1][c]), fmaxf(s_whi[2][c], s_whi[3][c]));
    }
    const double ex = x < blo[0] ? blo[0] - x : (x > bhi[0] ? x - bhi[0] : 0.0);
    const double ey = y < blo[1] ? blo[1] - y : (y > bhi[1] ? y - bhi[1] : 0.0);
    const double ez = z < blo[2] ? blo[2] - z : (z > bhi[2] ? z - bhi[2] : 0.0);
    const bool maybe = !found && sqrt(ex * ex + ey * ey + ez * ez) < thresh;   // box distance <= point distance
    if (__ballot(maybe) == 0ull) continue;
    if (maybe) {
      for (int t = 0; t < cnt; ++t) {
        const double dx = x - (double)s_r[t * 3], dy = y - (double)s_r[t * 3 + 1], dz = z - (double)s_r[t * 3 + 2];
        if (sqrt(dx * dx + dy * dy + dz * dz) < thresh) { found = true; break; }
      }
    }
  }
  if (live) near[i] = found ? 1 : 0;
}

extern "C" int sv_vcn_surface_select(const float* partial, const float* complete, int batch, int n_partial, int n_complete, int k,
                                     int surface_pts, float* surface, int32_t* n_selected, void* stream) {
  SV_CHECK_ARG(batch >= 0 && n_partial >= 1 && n_partial <= PP_MAXN && n_complete >= 1 && n_complete <= PP_MAXN,
               "sv_vcn_surface_select: 1 <= n_partial, n_complete <= %d (got %d, %d)", PP_MAXN, n_partial, n_complete);
  SV_CHECK_ARG(k >= 1 && k <= n_complete, "sv_vcn_surface_select: 1 <= k <= n_complete (k=%d)", k);
  SV_CHECK_ARG(surface_pts >= 1, "sv_vcn_surface_select: surface_pts >= 1");
  if (batch == 0) return SV_OK;
  SV_CHECK_ARG(partial && complete && surface && n_selected, "sv_vcn_surface_select: null pointer");
  SurfaceArgs a{partial, complete, surface, n_selected, n_partial, n_complete, k, surface_pts};
  hipLaunchKernelGGL(k_surface_select, dim3(batch), dim3(PP_THREADS), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_vcn_largest_cluster(const float* points, int batch, int n, double eps, int min_points, int total_pts, float* out,
                                      int32_t* n_cluster, void* stream) {
  SV_CHECK_ARG(batch >= 0 && n >= 1 && n <= PP_MAXN, "sv_vcn_largest_cluster: 1 <= n <= %d (got %d)", PP_MAXN, n);
  SV_CHECK_ARG(min_points >= 0 && min_points <= 2,
               "sv_vcn_largest_cluster: min_points <= 2 only (border-point order of open3d's BFS is not reproduced); got %d", min_points);
  SV_CHECK_ARG(total_pts >= 1 && eps > 0, "sv_vcn_largest_cluster: total_pts >= 1, eps > 0");
  if (batch == 0) return SV_OK;
  SV_CHECK_ARG(points && out && n_cluster, "sv_vcn_largest_cluster: null pointer");
  ClusterArgs a{points, out, n_cluster, n, total_pts, min_points, eps * eps};
  hipLaunchKernelGGL(k_largest_cluster, dim3(batch), dim3(PP_THREADS), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_points_near_set(const float* query, int64_t n_query, const float* ref, int64_t n_ref, double thresh, uint8_t* near,
                                  void* stream) {
  SV_CHECK_ARG(n_query >= 0 && n_ref >= 0 && thresh >= 0, "sv_points_near_set: negative size");
  if (n_query == 0) return SV_OK;
  SV_CHECK_ARG(query && near && (ref || n_ref == 0), "sv_points_near_set: null pointer");
  hipLaunchKernelGGL(k_points_near_set, dim3(sv_div_up(n_query, 256)), dim3(256), 0, sv_stream(stream), query, (long)n_query, ref,
                     (long)n_ref, thresh, near);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
