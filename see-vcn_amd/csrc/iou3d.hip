// Rotated-box geometry: BEV overlap / IoU for all pairs, rotated and axis-aligned NMS, points-in-boxes.
//
// Reference: detector3d/pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu (box_overlap :95-223, iou_bev :225-231,
// nms_kernel :267-302, nms_normal_kernel :317-355) + host sweep iou3d_nms.cpp:90-135;
// detector3d/pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-36,313-337 (points_in_boxes).
//
// What is different here (same results):
//   * a conservative bounding-circle test skips the polygon construction for pairs that cannot touch (after the score
//     sort of NMS that is almost every pair);
//   * NMS masks are produced one 64-bit word per wave-wide ballot and only for the upper triangle; the greedy sweep
//     runs ON THE GPU (one wave, 64 boxes per step: the diagonal word decides suppression inside the block, kept rows
//     are OR-ed into the remaining words with coalesced loads) — the reference copies the N x N/64 mask to the host
//     and sweeps there (iou3d_nms.cpp:111-131);
//   * per-box sin/cos are computed once per tile/box, not once per pair.
#include <math.h>

#include "common.h"

struct P2 { float x, y; };

__device__ __forceinline__ float crs(P2 p1, P2 p2, P2 p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

struct RBox {       // a box with its trigonometry evaluated once
  float x, y, dx, dy, ang, c, s;      // c = cos(ang), s = sin(ang)
  float nc, ns;                       // cos(-ang), sin(-ang) (separate evaluations, as the reference does)
};

__device__ __forceinline__ RBox make_rbox(const float* b) {
  RBox r;
  r.x = b[0]; r.y = b[1]; r.dx = b[3]; r.dy = b[4]; r.ang = b[6];
  r.c = cosf(r.ang); r.s = sinf(r.ang);
  r.nc = cosf(-r.ang); r.ns = sinf(-r.ang);
  return r;
}

__device__ __forceinline__ void corners(const RBox& b, P2 (&c)[5]) {
  const float hx = b.dx / 2, hy = b.dy / 2;
  const float px[4] = {b.x - hx, b.x + hx, b.x + hx, b.x - hx};
  const float py[4] = {b.y - hy, b.y - hy, b.y + hy, b.y + hy};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c[k].x = (px[k] - b.x) * b.c + (py[k] - b.y) * (-b.s) + b.x;
    c[k].y = (px[k] - b.x) * b.s + (py[k] - b.y) * b.c + b.y;
  }
  c[4] = c[0];
}

__device__ __forceinline__ bool inside(const RBox& b, P2 p) {   // with the reference's 1e-2 margin
  const float rx = (p.x - b.x) * b.nc + (p.y - b.y) * (-b.ns);
  const float ry = (p.x - b.x) * b.ns + (p.y - b.y) * b.nc;
  return fabsf(rx) < b.dx / 2 + 1e-2f && fabsf(ry) < b.dy / 2 + 1e-2f;
}

__device__ __forceinline__ bool seg_x(P2 p1, P2 p0, P2 q1, P2 q0, P2& ans) {
  const bool boxes_meet = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                          fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
  if (!boxes_meet) return false;
  const float s1 = crs(q0, p1, p0), s2 = crs(p1, q1, p0), s3 = crs(p0, q1, q0), s4 = crs(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = crs(q1, p1, p0);
  if (fabsf(s5 - s1) > 1e-8f) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

__device__ float overlap_area(const RBox& a, const RBox& b) {
  // pairs whose bounding circles (+ the 1e-2 corner margin, + slack) are apart have no crossing and no corner inside:
  // the construction below would return exactly 0
  {
    const float ra = 0.5f * sqrtf(a.dx * a.dx + a.dy * a.dy), rb = 0.5f * sqrtf(b.dx * b.dx + b.dy * b.dy);
    const float ddx = a.x - b.x, ddy = a.y - b.y, reach = ra + rb + 0.05f;
    if (ddx * ddx + ddy * ddy > reach * reach) return 0.f;
  }
  P2 A[5], B[5];
  corners(a, A);
  corners(b, B);
  P2 pts[24];
  float ang[24];
  int cnt = 0;
  float sx = 0.f, sy = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 x;
      if (seg_x(A[i + 1], A[i], B[j + 1], B[j], x)) { sx += x.x; sy += x.y; pts[cnt++] = x; }
    }
  for (int k = 0; k < 4; ++k) {
    if (inside(a, B[k])) { sx += B[k].x; sy += B[k].y; pts[cnt++] = B[k]; }
    if (inside(b, A[k])) { sx += A[k].x; sy += A[k].y; pts[cnt++] = A[k]; }
  }
  if (cnt < 3) return 0.f;                 // fewer than 3 points span no area (the reference's sums are empty or zero)
  sx /= cnt; sy /= cnt;
  for (int i = 0; i < cnt; ++i) ang[i] = atan2f(pts[i].y - sy, pts[i].x - sx);
  for (int j = 0; j < cnt - 1; ++j)        // same bubble order as the reference (stable for equal angles)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        const P2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
        const float u = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = u;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k)
    area += (pts[k].x - pts[0].x) * (pts[k + 1].y - pts[0].y) - (pts[k].y - pts[0].y) * (pts[k + 1].x - pts[0].x);
  return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const RBox& a, const RBox& b) {
  const float sa = a.dx * a.dy, sb = b.dx * b.dy, so = overlap_area(a, b);
  return so / fmaxf(sa + sb - so, 1e-8f);
}

__device__ __forceinline__ float iou_axis_aligned(const float* a, const float* b) {
  const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f), inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

// ------------------------------------------------------------------ all-pairs overlap / IoU: one thread per pair
__global__ __launch_bounds__(256) void k_boxes_pairs(int na, const float* __restrict__ A, int nb, const float* __restrict__ B,
                                                     float* __restrict__ out, int iou) {
  __shared__ RBox sb[64];
  // tile: 4 rows of A x 64 columns of B per workgroup
  const int col0 = blockIdx.x * 64, row0 = blockIdx.y * 4;
  if (threadIdx.x < 64 && col0 + threadIdx.x < nb) sb[threadIdx.x] = make_rbox(B + (int64_t)(col0 + threadIdx.x) * 7);
  __syncthreads();
  const int r = row0 + (threadIdx.x >> 6), c = col0 + (threadIdx.x & 63);
  if (r >= na || c >= nb) return;
  const RBox a = make_rbox(A + (int64_t)r * 7);
  const RBox& b = sb[threadIdx.x & 63];
  out[(int64_t)r * nb + c] = iou ? iou_bev(a, b) : overlap_area(a, b);
}

extern "C" int sv_boxes_overlap_bev(const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* out, int iou, void* stream) {
  SV_CHECK_ARG(num_a >= 0 && num_b >= 0, "boxes_overlap_bev: bad arguments");
  if (num_a == 0 || num_b == 0) return SV_OK;
  SV_CHECK_ARG(boxes_a && boxes_b && out, "boxes_overlap_bev: null pointer");
  dim3 grid(sv_div_up(num_b, 64), sv_div_up(num_a, 4));
  hipLaunchKernelGGL(k_boxes_pairs, grid, dim3(256), 0, sv_stream(stream), num_a, boxes_a, num_b, boxes_b, out, iou);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// 3-D IoU of every (a, b) pair of `batch` independent box sets: BEV overlap x height overlap / union volume, the arithmetic of the reference's
// boxes_iou3d_gpu (iou3d_nms_utils.py:48-81) operation by operation -- there ~22 elementwise launches around the overlap kernel, per scene.
// Boxes are rows of `stride` floats whose first 7 are [x,y,z,dx,dy,dz,heading] (ground-truth blocks carry a class column behind them).
__global__ __launch_bounds__(256) void k_boxes_iou3d(int na, const float* __restrict__ A, int stride_a, int nb, const float* __restrict__ B, int stride_b,
                                                     float* __restrict__ out) {
  __shared__ RBox sb[64];
  __shared__ float sbz[64][3];                                 // height max, height min, volume of the b boxes of the tile
  const int scene = blockIdx.z;
  A += (int64_t)scene * na * stride_a, B += (int64_t)scene * nb * stride_b, out += (int64_t)scene * na * nb;
  const int col0 = blockIdx.x * 64, row0 = blockIdx.y * 4;
  if (threadIdx.x < 64 && col0 + threadIdx.x < nb) {
    const float* b = B + (int64_t)(col0 + threadIdx.x) * stride_b;
    sb[threadIdx.x] = make_rbox(b);
    sbz[threadIdx.x][0] = b[2] + b[5] / 2.f, sbz[threadIdx.x][1] = b[2] - b[5] / 2.f, sbz[threadIdx.x][2] = b[3] * b[4] * b[5];
  }
  __syncthreads();
  const int r = row0 + (threadIdx.x >> 6), c = col0 + (threadIdx.x & 63);
  if (r >= na || c >= nb) return;
  const float* a = A + (int64_t)r * stride_a;
  const float a_hmax = a[2] + a[5] / 2.f, a_hmin = a[2] - a[5] / 2.f, vol_a = a[3] * a[4] * a[5];
  const float* bz = sbz[threadIdx.x & 63];
  const float overlap_bev = overlap_area(make_rbox(a), sb[threadIdx.x & 63]);
  const float overlap_h = fmaxf(fminf(a_hmax, bz[0]) - fmaxf(a_hmin, bz[1]), 0.f);
  const float overlap_3d = overlap_bev * overlap_h;
  out[(int64_t)r * nb + c] = overlap_3d / fmaxf(vol_a + bz[2] - overlap_3d, 1e-6f);
}

extern "C" int sv_boxes_iou3d_batch(const float* boxes_a, int num_a, int stride_a, const float* boxes_b, int num_b, int stride_b, int batch, float* out,
                                    void* stream) {
  SV_CHECK_ARG(num_a >= 0 && num_b >= 0 && batch >= 0 && stride_a >= 7 && stride_b >= 7, "boxes_iou3d_batch: bad arguments");
  if (num_a == 0 || num_b == 0 || batch == 0) return SV_OK;
  SV_CHECK_ARG(boxes_a && boxes_b && out && batch <= 65535, "boxes_iou3d_batch: null pointer or more than 65535 box sets");
  dim3 grid(sv_div_up(num_b, 64), sv_div_up(num_a, 4), batch);
  hipLaunchKernelGGL(k_boxes_iou3d, grid, dim3(256), 0, sv_stream(stream), num_a, boxes_a, stride_a, num_b, boxes_b, stride_b, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------ NMS: suppression masks (upper triangle)
// one wave per 64x64 tile; lane = column box; each row contributes one ballot word
// (row blocks rb0 + blockIdx.y of a chunked sweep; `done` non-null and set = an earlier chunk already kept max_keep boxes: nothing to do)
__global__ __launch_bounds__(64) void k_nms_mask(int n, float thresh, const float* __restrict__ boxes, unsigned long long* __restrict__ mask,
                                                 int col_blocks, int normal, int rb0, const int32_t* __restrict__ done) {
  if (done && *done) return;
  const int rb = rb0 + blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;                                       // the sweep never reads words left of the diagonal
  const int lane = threadIdx.x;
  const int col = cb * 64 + lane;
  const bool col_ok = col < n;
  float cbox[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) cbox[k] = col_ok ? boxes[(int64_t)col * 7 + k] : 0.f;
  const RBox c = make_rbox(cbox);
  const int rows = min(64, n - rb * 64);
  for (int r = 0; r < rows; ++r) {
    const int row = rb * 64 + r;
    float rbox[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) rbox[k] = boxes[(int64_t)row * 7 + k];   // wave-uniform address: one broadcast load
    bool sup = false;
    if (col_ok && col > row) sup = (normal ? iou_axis_aligned(rbox, cbox) : iou_bev(make_rbox(rbox), c)) > thresh;
    const unsigned long long w = __ballot(sup);
    if (lane == 0) mask[(int64_t)row * col_blocks + cb] = w;
  }
}

// greedy sweep on the device: one workgroup, removal bitmap in LDS (n <= 65536).  Per block of 64 boxes wave 0 resolves the block against
// itself (diagonal word of each row, 64 serial steps on registers), then all four waves fold the kept rows into the words of the later
// blocks (lanes over words: every row read is one coalesced run).  The sweep stops once max_keep boxes are kept: callers that only take the
// first NMS_POST_MAXSIZE survivors (model_nms_utils.py:21, roi_head_template.py:68-72) get the same prefix without the rest of the sweep
// (9000 proposals at threshold 0.8, 512 wanted: 4.3 ms -> 0.3 ms).
constexpr int NMS_MAX_WORDS = 1024;
constexpr int NMS_SWEEP_THREADS = 256;
// nb0 .. nb1: the blocks this launch resolves.  A whole sweep is one launch (nb0 = 0, nb1 = col_blocks, state = null); a CHUNKED sweep is
// a sequence of launches that carry the removal bitmap and the kept count in `state` ([0] kept so far, [1] done flag, then the bitmap
// words) -- the mask rows of a chunk are computed right before its launch and not at all once max_keep is reached (k_nms_mask checks the
// flag): 9000 proposals at threshold 0.8 keep their 512th box inside the first 1024, the other 88 % of the 40 M rotated IoUs are never made.
struct NmsState {
  int32_t kept, done;
  unsigned long long remv[NMS_MAX_WORDS];
};
__global__ __launch_bounds__(NMS_SWEEP_THREADS) void k_nms_sweep(int n, const unsigned long long* __restrict__ mask, int col_blocks, int max_keep,
                                                                 int64_t* __restrict__ keep, int32_t* __restrict__ num_out, int nb0, int nb1,
                                                                 NmsState* __restrict__ state) {
  __shared__ unsigned long long remv[NMS_MAX_WORDS];
  __shared__ unsigned long long s_kept;
  const int tid = threadIdx.x, lane = tid & 63;
  if (state && state->done) return;                            // (the count was written by the launch that set the flag)
  for (int j = tid; j < col_blocks; j += NMS_SWEEP_THREADS) remv[j] = (state && nb0 > 0) ? state->remv[j] : 0ull;
  __syncthreads();
  int kept_total = (state && nb0 > 0) ? state->kept : 0;
  for (int nb = nb0; nb < nb1 && kept_total < max_keep; ++nb) {
    if (tid < 64) {
      const int rows = min(64, n - nb * 64);
      unsigned long long cur = remv[nb];
      const unsigned long long diag = lane < rows ? mask[(int64_t)(nb * 64 + lane) * col_blocks + nb] : 0ull;
      unsigned long long keptbits = 0ull;
      for (int r = 0; r < rows; ++r) {
        const unsigned long long d = __shfl(diag, r, 64);
        if (!((cur >> r) & 1ull)) {
          keptbits |= 1ull << r;
          cur |= d;
        }
      }
      const int slot = kept_total + __popcll(keptbits & ((1ull << lane) - 1ull));
      if (((keptbits >> lane) & 1ull) && slot < max_keep) keep[slot] = (int64_t)nb * 64 + lane;
      if (lane == 0) s_kept = keptbits;
    }
    __syncthreads();
    const unsigned long long keptbits = s_kept;
    kept_total += __popcll(keptbits);
    if (kept_total < max_keep) {
      for (int j = nb + 1 + tid; j < col_blocks; j += NMS_SWEEP_THREADS) {
        unsigned long long acc = remv[j];
        unsigned long long kb = keptbits;
        const unsigned long long* col = mask + (int64_t)nb * 64 * col_blocks + j;
        while (kb) {
          const int r = __ffsll((long long)kb) - 1;
          kb &= kb - 1;
          acc |= col[(int64_t)r * col_blocks];
        }
        remv[j] = acc;
      }
    }
    __syncthreads();
  }
  const bool finished = kept_total >= max_keep || nb1 >= col_blocks;
  if (state) {
    if (!finished)
      for (int j = tid; j < col_blocks; j += NMS_SWEEP_THREADS) state->remv[j] = remv[j];
    if (tid == 0) state->kept = kept_total, state->done = finished ? 1 : 0;
  }
  if (tid == 0 && finished) *num_out = min(kept_total, max_keep);
}

extern "C" size_t sv_nms_scratch_bytes(int n) {
  const size_t cb = (size_t)(n + 63) / 64;
  return ((size_t)n * cb + 8) * sizeof(unsigned long long) + sizeof(NmsState);
}

extern "C" int sv_nms_prefix(const float* boxes, int n, float thresh, int normal, int max_keep, void* scratch, int64_t* keep, int32_t* num_out,
                             void* stream);
extern "C" int sv_nms(const float* boxes, int n, float thresh, int normal, void* scratch, int64_t* keep, int32_t* num_out, void* stream) {
  return sv_nms_prefix(boxes, n, thresh, normal, n, scratch, keep, num_out, stream);
}

extern "C" int sv_nms_prefix(const float* boxes, int n, float thresh, int normal, int max_keep, void* scratch, int64_t* keep, int32_t* num_out,
                             void* stream) {
  SV_CHECK_ARG(n >= 0 && num_out && max_keep >= 0, "nms: bad arguments");
  hipStream_t st = sv_stream(stream);
  if (n == 0) {
    SV_HIP(hipMemsetAsync(num_out, 0, 4, st));
    return SV_OK;
  }
  SV_CHECK_ARG(boxes && scratch && keep, "nms: null pointer");
  const int cb = (n + 63) / 64;
  SV_CHECK_ARG(cb <= NMS_MAX_WORDS, "nms: at most %d boxes", NMS_MAX_WORDS * 64);
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(scratch);
  if (max_keep >= n || cb <= 32) {                             // the whole sweep is wanted, or it is small: two launches
    hipLaunchKernelGGL(k_nms_mask, dim3(cb, cb), dim3(64), 0, st, n, thresh, boxes, mask, cb, normal, 0, (const int32_t*)nullptr);
    hipLaunchKernelGGL(k_nms_sweep, dim3(1), dim3(NMS_SWEEP_THREADS), 0, st, n, mask, cb, max_keep, keep, num_out, 0, cb, (NmsState*)nullptr);
    SV_LAUNCH_CHECK();
    return SV_OK;
  }
  // a prefix is wanted: row chunks of growing size (16, 32, 64, ... blocks), masks of a chunk only if the chunks before did not reach max_keep
  NmsState* state = reinterpret_cast<NmsState*>(mask + (size_t)n * cb + 8);
  SV_HIP(hipMemsetAsync(state, 0, 8, st));
  for (int nb0 = 0, step = 16; nb0 < cb; nb0 += step, step *= 2) {
    const int nb1 = nb0 + step < cb ? nb0 + step : cb;
    hipLaunchKernelGGL(k_nms_mask, dim3(cb, nb1 - nb0), dim3(64), 0, st, n, thresh, boxes, mask, cb, normal, nb0, &state->done);
    hipLaunchKernelGGL(k_nms_sweep, dim3(1), dim3(NMS_SWEEP_THREADS), 0, st, n, mask, cb, max_keep, keep, num_out, nb0, nb1, state);
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------ points in boxes
// boxes (B,T,7), pts (B,M,3) -> idx (B,M): first box containing the point, else -1
__global__ __launch_bounds__(256) void k_points_in_boxes(int T, int M, const float* __restrict__ boxes, const float* __restrict__ pts,
                                                         int32_t* __restrict__ out) {
  extern __shared__ float sbox[];                            // [T][9]: cx cy cz dx dy dz cos(-rz) sin(-rz)
  const int b = blockIdx.y;
  for (int k = threadIdx.x; k < T; k += blockDim.x) {
    const float* bx = boxes + ((int64_t)b * T + k) * 7;
    float* s = sbox + k * 8;
    s[0] = bx[0]; s[1] = bx[1]; s[2] = bx[2]; s[3] = bx[3]; s[4] = bx[4]; s[5] = bx[5];
    s[6] = cosf(-bx[6]); s[7] = sinf(-bx[6]);
  }
  __syncthreads();
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float* p = pts + ((int64_t)b * M + m) * 3;
  const float x = p[0], y = p[1], z = p[2];
  int32_t found = -1;
  for (int k = 0; k < T; ++k) {
    const float* s = sbox + k * 8;
    if (fabsf(z - s[2]) > s[5] / 2.0) continue;              // double-precision compare like the reference (:28)
    const float sx = x - s[0], sy = y - s[1];
    const float lx = sx * s[6] + sy * (-s[7]), ly = sx * s[7] + sy * s[6];
    if (fabs(lx) < s[3] / 2.0 + (double)1e-5f && fabs(ly) < s[4] / 2.0 + (double)1e-5f) { found = k; break; }
  }
  out[(int64_t)b * M + m] = found;
}

extern "C" int sv_points_in_boxes(const float* boxes, const float* pts, int batch, int num_boxes, int num_points, int32_t* out, void* stream) {
  SV_CHECK_ARG(batch >= 0 && num_boxes >= 0 && num_points >= 0, "points_in_boxes: bad arguments");
  if (batch == 0 || num_points == 0) return SV_OK;
  SV_CHECK_ARG(pts && out && (num_boxes == 0 || boxes), "points_in_boxes: null pointer");
  SV_CHECK_ARG((size_t)num_boxes * 32 <= 64 * 1024, "points_in_boxes: more than 2048 boxes per scene");
  dim3 grid(sv_div_up(num_points, 256), batch);
  hipLaunchKernelGGL(k_points_in_boxes, grid, dim3(256), (size_t)num_boxes * 32, sv_stream(stream), num_boxes, num_points, boxes, pts, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// points_in_boxes_cpu (roiaware_pool3d.cpp:121-165): (N boxes x M points) 0/1 matrix, MARGIN 1e-2 (the GPU variant uses 1e-5)
__global__ __launch_bounds__(256) void k_points_in_boxes_matrix(int N, int M, const float* __restrict__ boxes, const float* __restrict__ pts,
                                                                int32_t* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= M) return;
  const float* b = boxes + (size_t)i * 7;
  const float x = pts[(size_t)j * 3], y = pts[(size_t)j * 3 + 1], z = pts[(size_t)j * 3 + 2];
  int in = 0;
  if (!(fabsf(z - b[2]) > b[5] / 2.0)) {
    const float cosa = cosf(-b[6]), sina = sinf(-b[6]);
    const float sx = x - b[0], sy = y - b[1];
    const float lx = sx * cosa + sy * (-sina), ly = sx * sina + sy * cosa;
    in = (fabs(lx) < b[3] / 2.0 + (double)1e-2f && fabs(ly) < b[4] / 2.0 + (double)1e-2f) ? 1 : 0;
  }
  out[(size_t)i * M + j] = in;
}

extern "C" int sv_points_in_boxes_matrix(const float* boxes, const float* pts, int num_boxes, int num_points, int32_t* out, void* stream) {
  SV_CHECK_ARG(num_boxes >= 0 && num_points >= 0, "sv_points_in_boxes_matrix: negative size");
  if (num_boxes == 0 || num_points == 0) return SV_OK;
  SV_CHECK_ARG(boxes && pts && out, "sv_points_in_boxes_matrix: null pointer");
  hipLaunchKernelGGL(k_points_in_boxes_matrix, dim3(sv_div_up(num_points, 256), num_boxes), dim3(256), 0, sv_stream(stream), num_boxes,
                     num_points, boxes, pts, out);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
