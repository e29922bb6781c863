// BatchNorm1d (+ ReLU) over sparse-voxel feature matrices (N, C), channel-last — the `norm_fn -> ReLU` tail of every
// post_act_block (detector3d/pcdet/models/backbones_3d/spconv_backbone.py:9-27,73: BatchNorm1d eps 1e-3, momentum 0.01).
// HBM-bound: forward = read x twice + write y, backward = read x,dy twice + write dx.  Three launches each way: row-chunk
// partial sums, a per-channel fp64 combine in a fixed order (deterministic), one elementwise pass.
// (Round 2 tried twice to let the workgroup that arrives last do the combine inside the reduce launch: with a release fence per workgroup
// it took 67-71 us instead of 12 + 5; with write-through (sc1) partials, a drained arrival counter and an acquire in the last arriver only
// it took 30-34 us -- one workgroup reading 1024 x 2 x C partials from memory is slower than C workgroups doing it in a launch of their
// own -- and one test saw a stale partial at 4 workgroups per CU.  A kernel boundary costs 1.5 us here; three launches it is.)
#include "common.h"

#include "norm.h"

// per-thread: channels c4*4..c4*4+3 of rows (row0 + tid / C4) + k * R
template <bool BWD>
__global__ __launch_bounds__(BN_THREADS) void k_bn_reduce(BnArgs a) {
  __shared__ float s_red[2][BN_THREADS * 4];
  const int tid = threadIdx.x, C4 = a.C >> 2;
  const int R = BN_THREADS / C4;                       // rows per sweep (C4 divides 256 for C in {4..512} powers of two; else idle lanes)
  const int c4 = tid % C4, rr = tid / C4;
  const bool active = rr < R;
  const int64_t rows_per_wg = (a.n + a.wgs - 1) / a.wgs;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = min(a.n, r0 + rows_per_wg);
  float4 s0 = make_float4(0, 0, 0, 0), s1 = make_float4(0, 0, 0, 0);
  float4 mean = make_float4(0, 0, 0, 0), istd = mean, g = make_float4(1, 1, 1, 1), bt = mean;
  if (BWD && active) {
    mean = reinterpret_cast<const float4*>(a.save_mean)[c4], istd = reinterpret_cast<const float4*>(a.save_invstd)[c4];
    if (a.gamma) g = reinterpret_cast<const float4*>(a.gamma)[c4];
    if (a.beta) bt = reinterpret_cast<const float4*>(a.beta)[c4];
  }
  const float4 sc = make_float4(istd.x * g.x, istd.y * g.y, istd.z * g.z, istd.w * g.w);
  const float4 sh = make_float4(bn_shift(mean.x, sc.x, bt.x), bn_shift(mean.y, sc.y, bt.y), bn_shift(mean.z, sc.z, bt.z), bn_shift(mean.w, sc.w, bt.w));
  if (active) {
    for (int64_t r = r0 + rr; r < r1; r += R) {
      const float4 v = reinterpret_cast<const float4*>(a.x + r * a.C)[c4];
      if (!BWD) {
        s0.x += v.x, s0.y += v.y, s0.z += v.z, s0.w += v.w;
        s1.x += v.x * v.x, s1.y += v.y * v.y, s1.z += v.z * v.z, s1.w += v.w * v.w;
      } else {
        float4 d = reinterpret_cast<const float4*>(a.dy + r * a.C)[c4];
        const float4 xh = make_float4((v.x - mean.x) * istd.x, (v.y - mean.y) * istd.y, (v.z - mean.z) * istd.z, (v.w - mean.w) * istd.w);
        if (a.relu) {   // ReLU branch recomputed from x with the forward's own expression
          d.x = (bn_act(v.x, sc.x, sh.x) > 0.f) ? d.x : 0.f, d.y = (bn_act(v.y, sc.y, sh.y) > 0.f) ? d.y : 0.f;
          d.z = (bn_act(v.z, sc.z, sh.z) > 0.f) ? d.z : 0.f, d.w = (bn_act(v.w, sc.w, sh.w) > 0.f) ? d.w : 0.f;
        }
        s0.x += d.x, s0.y += d.y, s0.z += d.z, s0.w += d.w;
        s1.x += d.x * xh.x, s1.y += d.y * xh.y, s1.z += d.z * xh.z, s1.w += d.w * xh.w;
      }
    }
  }
  reinterpret_cast<float4*>(s_red[0])[tid] = s0;
  reinterpret_cast<float4*>(s_red[1])[tid] = s1;
  __syncthreads();
  if (tid < C4) {       // fixed-order sum over the R row groups
    float4 t0 = make_float4(0, 0, 0, 0), t1 = t0;
    for (int q = 0; q < R; ++q) {
      const float4 u0 = reinterpret_cast<float4*>(s_red[0])[q * C4 + tid], u1 = reinterpret_cast<float4*>(s_red[1])[q * C4 + tid];
      t0.x += u0.x, t0.y += u0.y, t0.z += u0.z, t0.w += u0.w;
      t1.x += u1.x, t1.y += u1.y, t1.z += u1.z, t1.w += u1.w;
    }
    float* p = a.partial + (size_t)blockIdx.x * 2 * a.C;
    reinterpret_cast<float4*>(p)[tid] = t0;
    reinterpret_cast<float4*>(p + a.C)[tid] = t1;
  }
}

// per-channel statistics from the workgroup partials: one workgroup per channel, fp64 tree in a fixed order
// (tried: one workgroup per 8 channels with every thread walking each 32nd partial row of one channel -- coalesced 32-byte pieces instead of one
// 128-byte line per thread and row -- 12.5 / 9.9 us instead of 5.7 / 5.0: eight workgroups with 32 dependent loads per thread lose to 64 x 4;
// round 5: ONE wave per channel, its <= 16 partial pairs per lane requested up front and the lane sums met by shuffles instead of the eight-barrier LDS
// tree -- the step went 3.64 -> 3.78 ms (profiles/r05_fin_ab.txt): 32 load instructions of 64 different lines each through one wave's address unit take
// longer than the four waves' shorter queues.  Measured with SEEVCN_DEBUG_SKIP_FINALIZE: the 12 forward launches cost the step 0.08 ms in all.)
template <bool BWD>
__global__ __launch_bounds__(BN_THREADS) void k_bn_finalize(BnArgs a) {
  __shared__ double s0[BN_THREADS], s1[BN_THREADS];
  const int c = blockIdx.x, tid = threadIdx.x;
  if (!BWD && c == 0 && tid == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
  double t0 = 0.0, t1 = 0.0;
  for (int w = tid; w < a.wgs; w += BN_THREADS) {
    t0 += (double)a.partial[(size_t)w * 2 * a.C + c];
    t1 += (double)a.partial[(size_t)w * 2 * a.C + a.C + c];
  }
  s0[tid] = t0, s1[tid] = t1;
  __syncthreads();
  for (int off = BN_THREADS / 2; off > 0; off >>= 1) {
    if (tid < off) s0[tid] += s0[tid + off], s1[tid] += s1[tid + off];
    __syncthreads();
  }
  if (tid != 0) return;
  t0 = s0[0], t1 = s1[0];
  const float gm = a.gamma ? a.gamma[c] : 1.f, bb = a.beta ? a.beta[c] : 0.f;
  if (!BWD) {
    const double m = t0 / (double)a.n;
    double var = t1 / (double)a.n - m * m;             // biased variance of the batch
    var = var < 0.0 ? 0.0 : var;
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    a.save_mean[c] = (float)m, a.save_invstd[c] = invstd;
    if (a.running_mean) {
      const double unbiased = a.n > 1 ? var * (double)a.n / (double)(a.n - 1) : var;
      a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)m;
      a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
    }
    const float scale = invstd * gm;                    // y = x * scale + shift
    a.coef[c] = scale;
    a.coef[a.C + c] = bn_shift((float)m, scale, bb);
  } else {
    const float invstd = a.save_invstd[c], m = a.save_mean[c];
    a.dbeta[c] = (float)t0, a.dgamma[c] = (float)t1;
    // dx = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat)),  xhat = (x - m) * invstd
    a.coef[c] = gm * invstd;
    a.coef[a.C + c] = (float)(t0 / (double)a.n);
    a.coef[2 * a.C + c] = (float)(t1 / (double)a.n);
    a.coef[3 * a.C + c] = m;
  }
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_apply_fwd(BnArgs a) {
  const int C4 = a.C >> 2;
  const int64_t total = a.n * C4;
  for (int64_t i = (int64_t)blockIdx.x * BN_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * BN_THREADS) {
    const int c4 = (int)(i % C4);
    const float4 v = reinterpret_cast<const float4*>(a.x)[i];
    const float4 sc = reinterpret_cast<const float4*>(a.coef)[c4], sh = reinterpret_cast<const float4*>(a.coef + a.C)[c4];
    float4 y = make_float4(bn_act(v.x, sc.x, sh.x), bn_act(v.y, sc.y, sh.y), bn_act(v.z, sc.z, sh.z), bn_act(v.w, sc.w, sh.w));
    if (a.relu) y.x = fmaxf(y.x, 0.f), y.y = fmaxf(y.y, 0.f), y.z = fmaxf(y.z, 0.f), y.w = fmaxf(y.w, 0.f);
    reinterpret_cast<float4*>(a.out)[i] = y;
  }
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_apply_bwd(BnArgs a) {
  const int C4 = a.C >> 2;
  const int64_t total = a.n * C4;
  for (int64_t i = (int64_t)blockIdx.x * BN_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * BN_THREADS) {
    const int c4 = (int)(i % C4);
    const float4 v = reinterpret_cast<const float4*>(a.x)[i];
    float4 d = reinterpret_cast<const float4*>(a.dy)[i];
    const float4 k = reinterpret_cast<const float4*>(a.coef)[c4], md = reinterpret_cast<const float4*>(a.coef + a.C)[c4];
    const float4 mx = reinterpret_cast<const float4*>(a.coef + 2 * a.C)[c4], m = reinterpret_cast<const float4*>(a.coef + 3 * a.C)[c4];
    const float4 istd = reinterpret_cast<const float4*>(a.save_invstd)[c4];
    const float4 xh = make_float4((v.x - m.x) * istd.x, (v.y - m.y) * istd.y, (v.z - m.z) * istd.z, (v.w - m.w) * istd.w);
    if (a.relu) {
      const float4 g = a.gamma ? reinterpret_cast<const float4*>(a.gamma)[c4] : make_float4(1, 1, 1, 1);
      const float4 bt = a.beta ? reinterpret_cast<const float4*>(a.beta)[c4] : make_float4(0, 0, 0, 0);
      const float4 sc = make_float4(istd.x * g.x, istd.y * g.y, istd.z * g.z, istd.w * g.w);       // the forward's scale / shift, bit for bit
      const float4 sh = make_float4(bn_shift(m.x, sc.x, bt.x), bn_shift(m.y, sc.y, bt.y), bn_shift(m.z, sc.z, bt.z), bn_shift(m.w, sc.w, bt.w));
      d.x = (bn_act(v.x, sc.x, sh.x) > 0.f) ? d.x : 0.f, d.y = (bn_act(v.y, sc.y, sh.y) > 0.f) ? d.y : 0.f;
      d.z = (bn_act(v.z, sc.z, sh.z) > 0.f) ? d.z : 0.f, d.w = (bn_act(v.w, sc.w, sh.w) > 0.f) ? d.w : 0.f;
    }
    reinterpret_cast<float4*>(a.out)[i] = make_float4(k.x * (d.x - md.x - xh.x * mx.x), k.y * (d.y - md.y - xh.y * mx.y),
                                                      k.z * (d.z - md.z - xh.z * mx.z), k.w * (d.w - md.w - xh.w * mx.w));
  }
}

// Backward combine + elementwise pass in ONE launch (round 6; SEEVCN_BN_BWD_FUSED=1, OFF by default: measured SLOWER -- the step 3.56 / 3.58 ms with the
// two launches, 3.63 / 3.64 ms with this one, two same-box alternations, profiles/r06_bnfused_ab.txt: 256 workgroups each pulling the 512 KB of partials out
// of L2 beside the resident weight-gradient / data-gradient launches cost more than the small kernel and its boundary).  k_bn_finalize<true> is 64-128 small workgroups whose result every workgroup of
// k_bn_apply_bwd needs: on the backward chain of the benchmarked step that pair cost 20.8 + 19.7 us per BatchNorm (12 of them, in-step durations,
// profiles/r06_a_main_steady_kernels.csv; 5 + 13 us with the GPU otherwise idle) plus a kernel boundary.  Here every workgroup (one per CU, 1024 threads)
// first combines ALL partial sums itself -- (wgs, 2, C) floats, 512 KB at C = 64 out of L2: thread (g, quad) adds the rows g, g + G, ... of its four
// columns in fp64, the G group sums of a column are added in group order -- every workgroup the same values in the same order, so the coefficients are
// identical everywhere and reproducible; workgroup 0 also writes dgamma / dbeta.  Then the elementwise pass with the coefficients in LDS.
// (The combine's ORDER differs from k_bn_finalize<true>'s strided tree: both conv-side callers -- sv_batchnorm_relu_backward and _partial -- go through
// this kernel, so the launch-list chain and the per-module path still agree bit for bit.)
constexpr int BNF_THREADS = 1024;
__global__ __launch_bounds__(BNF_THREADS) void k_bn_bwd_fused(BnArgs a) {
  __shared__ double s_sum[BNF_THREADS * 4];                 // [group][2 C] column sums of a row group (G * 2 C = 4096 doubles whatever C is)
  __shared__ __attribute__((aligned(16))) float s_coef[4 * BN_MAX_C];       // gamma * invstd | mean(dy) | mean(dy * xhat) | mean
  const int tid = threadIdx.x, C = a.C, C2 = 2 * C, Q = C2 / 4, G = BNF_THREADS / Q;
  {
    const int q = tid % Q, g = tid / Q;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    if (g < G) {
      for (int w = g; w < a.wgs; w += G) {
        const float4 v = reinterpret_cast<const float4*>(a.partial + (size_t)w * C2)[q];
        t0 += (double)v.x, t1 += (double)v.y, t2 += (double)v.z, t3 += (double)v.w;
      }
      double* d = s_sum + (size_t)g * C2 + 4 * q;
      d[0] = t0, d[1] = t1, d[2] = t2, d[3] = t3;
    }
  }
  __syncthreads();
  if (tid < C) {
    double t0 = 0.0, t1 = 0.0;
    for (int g = 0; g < G; ++g) t0 += s_sum[(size_t)g * C2 + tid], t1 += s_sum[(size_t)g * C2 + C + tid];
    const float gm = a.gamma ? a.gamma[tid] : 1.f;
    const float invstd = a.save_invstd[tid], m = a.save_mean[tid];
    if (blockIdx.x == 0) a.dbeta[tid] = (float)t0, a.dgamma[tid] = (float)t1;
    // dx = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat)),  xhat = (x - m) * invstd      (k_bn_finalize<true>'s coefficients)
    s_coef[tid] = gm * invstd;
    s_coef[C + tid] = (float)(t0 / (double)a.n);
    s_coef[2 * C + tid] = (float)(t1 / (double)a.n);
    s_coef[3 * C + tid] = m;
  }
  __syncthreads();
  const int C4 = C >> 2;
  const int64_t total = a.n * C4;
  for (int64_t i = (int64_t)blockIdx.x * BNF_THREADS + tid; i < total; i += (int64_t)gridDim.x * BNF_THREADS) {
    const int c4 = (int)(i % C4);
    const float4 v = reinterpret_cast<const float4*>(a.x)[i];
    float4 d = reinterpret_cast<const float4*>(a.dy)[i];
    const float4 k = reinterpret_cast<const float4*>(s_coef)[c4], md = reinterpret_cast<const float4*>(s_coef + C)[c4];
    const float4 mx = reinterpret_cast<const float4*>(s_coef + 2 * C)[c4], m = reinterpret_cast<const float4*>(s_coef + 3 * C)[c4];
    const float4 istd = reinterpret_cast<const float4*>(a.save_invstd)[c4];
    const float4 xh = make_float4((v.x - m.x) * istd.x, (v.y - m.y) * istd.y, (v.z - m.z) * istd.z, (v.w - m.w) * istd.w);
    if (a.relu) {
      const float4 g = a.gamma ? reinterpret_cast<const float4*>(a.gamma)[c4] : make_float4(1, 1, 1, 1);
      const float4 bt = a.beta ? reinterpret_cast<const float4*>(a.beta)[c4] : make_float4(0, 0, 0, 0);
      const float4 sc = make_float4(istd.x * g.x, istd.y * g.y, istd.z * g.z, istd.w * g.w);       // the forward's scale / shift, bit for bit
      const float4 sh = make_float4(bn_shift(m.x, sc.x, bt.x), bn_shift(m.y, sc.y, bt.y), bn_shift(m.z, sc.z, bt.z), bn_shift(m.w, sc.w, bt.w));
      d.x = (bn_act(v.x, sc.x, sh.x) > 0.f) ? d.x : 0.f, d.y = (bn_act(v.y, sc.y, sh.y) > 0.f) ? d.y : 0.f;
      d.z = (bn_act(v.z, sc.z, sh.z) > 0.f) ? d.z : 0.f, d.w = (bn_act(v.w, sc.w, sh.w) > 0.f) ? d.w : 0.f;
    }
    reinterpret_cast<float4*>(a.out)[i] = make_float4(k.x * (d.x - md.x - xh.x * mx.x), k.y * (d.y - md.y - xh.y * mx.y),
                                                      k.z * (d.z - md.z - xh.z * mx.z), k.w * (d.w - md.w - xh.w * mx.w));
  }
}
// combine + elementwise pass of a BatchNorm backward whose partial sums are in a.partial: k_bn_finalize<true> + k_bn_apply_bwd, or (SEEVCN_BN_BWD_FUSED=1)
// the fused launch
static void bn_backward_tail(const BnArgs& a, hipStream_t st) {
  static const int fused = getenv("SEEVCN_BN_BWD_FUSED") ? atoi(getenv("SEEVCN_BN_BWD_FUSED")) : 0;
  const int Q = a.C / 2;
  if (fused && Q >= 1 && BNF_THREADS % Q == 0) {
    const int64_t quads = a.n * (a.C / 4);
    int wgs = (int)((quads + BNF_THREADS * 4 - 1) / (BNF_THREADS * 4));               // at least four sweeps per workgroup ...
    if (wgs > 256) wgs = 256;                                                         // ... and at most one workgroup per CU: each pays the combine
    if (wgs < 1) wgs = 1;
    hipLaunchKernelGGL(k_bn_bwd_fused, dim3(wgs), dim3(BNF_THREADS), 0, st, a);
    return;
  }
  hipLaunchKernelGGL(k_bn_finalize<true>, dim3(a.C), dim3(BN_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_bn_apply_bwd, dim3(sv_grid_1d(a.n * (a.C / 4), BN_THREADS)), dim3(BN_THREADS), 0, st, a);
}

// eval mode: per-channel scale/shift from the running statistics (one tiny launch), then the same apply kernel
__global__ void k_bn_eval_coef(BnArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.C) return;
  const float invstd = 1.f / sqrtf(a.running_var[c] + a.eps);
  const float gm = a.gamma ? a.gamma[c] : 1.f, bb = a.beta ? a.beta[c] : 0.f;
  a.coef[c] = invstd * gm;
  a.coef[a.C + c] = bb - a.running_mean[c] * invstd * gm;
}

static int bn_wgs(int64_t n, int C) {
  const int R = BN_THREADS / (C / 4);
  int64_t w = (n + (int64_t)R * 8 - 1) / ((int64_t)R * 8);   // at least 8 sweeps per workgroup
  if (w < 1) w = 1;
  if (w > BN_MAX_WGS) w = BN_MAX_WGS;
  return (int)w;
}

extern "C" size_t sv_batchnorm_scratch_bytes(int channels) {
  return ((size_t)BN_MAX_WGS * 2 * channels + 4 * (size_t)channels) * sizeof(float);
}

static int bn_common_check(const char* who, int64_t n, int C) {
  SV_CHECK_ARG(n >= 1, "%s: needs at least one row (got %ld)", who, (long)n);
  SV_CHECK_ARG(C >= 4 && C <= BN_MAX_C && (C & 3) == 0 && BN_THREADS % (C / 4) == 0,
               "%s: channels must be a multiple of 4 with C/4 dividing %d, <= %d (got %d)", who, BN_THREADS, BN_MAX_C, C);
  return SV_OK;
}

static void bn_scratch(BnArgs& a, void* scratch) {
  a.coef = reinterpret_cast<float*>(scratch);
  a.partial = a.coef + 4 * (size_t)a.C;
}

extern "C" int sv_batchnorm_relu_forward(const float* x, int64_t n, int channels, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float momentum, float eps, int training, int relu,
                                         void* scratch, float* y, float* save_mean, float* save_invstd, int64_t* num_batches_tracked,
                                         void* stream) {
  if (int rc = bn_common_check("sv_batchnorm_relu_forward", n, channels)) return rc;
  SV_CHECK_ARG(x && y && scratch, "sv_batchnorm_relu_forward: null pointer");
  SV_CHECK_ARG(training ? (save_mean && save_invstd) : (running_mean && running_var),
               "sv_batchnorm_relu_forward: training needs save_mean/save_invstd, eval needs running statistics");
  BnArgs a{};
  a.x = x, a.out = y, a.gamma = gamma, a.beta = beta, a.running_mean = running_mean, a.running_var = running_var;
  a.save_mean = save_mean, a.save_invstd = save_invstd, a.n = n, a.C = channels, a.relu = relu, a.momentum = momentum, a.eps = eps;
  a.wgs = bn_wgs(n, channels);
  a.num_batches_tracked = training ? num_batches_tracked : nullptr;
  bn_scratch(a, scratch);
  hipStream_t st = sv_stream(stream);
  if (training) {
    hipLaunchKernelGGL(k_bn_reduce<false>, dim3(a.wgs), dim3(BN_THREADS), 0, st, a);
    hipLaunchKernelGGL(k_bn_finalize<false>, dim3(channels), dim3(BN_THREADS), 0, st, a);
  } else
    hipLaunchKernelGGL(k_bn_eval_coef, dim3(sv_div_up(channels, 128)), dim3(128), 0, st, a);
  hipLaunchKernelGGL(k_bn_apply_fwd, dim3(sv_grid_1d(n * (channels / 4), BN_THREADS)), dim3(BN_THREADS), 0, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// The same training-mode forward with the statistics' first pass already done: `scratch` (laid out as for sv_batchnorm_relu_forward) holds
// n_partials per-workgroup partial sums (n_partials, 2, C) behind its 4 * C coefficient floats -- written there by the epilogue of the kernel
// that produced x (sv_sparse_conv_gather_gemm_planned's bn_partial).  Two launches instead of three, and x is read once instead of twice.
extern "C" int sv_batchnorm_relu_forward_partial(const float* x, int64_t n, int channels, const float* gamma, const float* beta, float* running_mean,
                                                 float* running_var, float momentum, float eps, int relu, void* scratch, int n_partials, float* y,
                                                 float* save_mean, float* save_invstd, int64_t* num_batches_tracked, void* stream) {
  if (int rc = bn_common_check("sv_batchnorm_relu_forward_partial", n, channels)) return rc;
  SV_CHECK_ARG(x && y && scratch && save_mean && save_invstd, "sv_batchnorm_relu_forward_partial: null pointer");
  SV_CHECK_ARG(n_partials >= 1 && n_partials <= BN_MAX_WGS, "sv_batchnorm_relu_forward_partial: 1..%d partials (got %d)", BN_MAX_WGS, n_partials);
  BnArgs a{};
  a.x = x, a.out = y, a.gamma = gamma, a.beta = beta, a.running_mean = running_mean, a.running_var = running_var;
  a.save_mean = save_mean, a.save_invstd = save_invstd, a.n = n, a.C = channels, a.relu = relu, a.momentum = momentum, a.eps = eps;
  a.wgs = n_partials;
  a.num_batches_tracked = num_batches_tracked;
  bn_scratch(a, scratch);
  hipStream_t st = sv_stream(stream);
  hipLaunchKernelGGL(k_bn_finalize<false>, dim3(channels), dim3(BN_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_bn_apply_fwd, dim3(sv_grid_1d(n * (channels / 4), BN_THREADS)), dim3(BN_THREADS), 0, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// The training-mode statistics WITHOUT the elementwise pass: from n_partials per-workgroup partial sums in `scratch` (as for
// sv_batchnorm_relu_forward_partial) to save_mean / save_invstd, the running statistics, and coef (2, C) = scale | shift of y = x * scale + shift in the
// CALLER's buffer (the shared scratch is overwritten by the next norm of the same width).  The consumers of y apply the coefficients as they read x
// (sv_conv_next_input_norm), or sv_batchnorm_apply materialises y where a tensor is needed.  One launch; x != NULL: the producer left no partials, the
// statistics pass over x (n, C) runs first (n_partials ignored).
extern "C" int sv_batchnorm_finalize_forward(const float* x, int64_t n, int channels, const float* gamma, const float* beta, float* running_mean,
                                             float* running_var, float momentum, float eps, void* scratch, int n_partials, float* coef, float* save_mean,
                                             float* save_invstd, int64_t* num_batches_tracked, void* stream) {
  if (int rc = bn_common_check("sv_batchnorm_finalize_forward", n, channels)) return rc;
  SV_CHECK_ARG(scratch && coef && save_mean && save_invstd, "sv_batchnorm_finalize_forward: null pointer");
  SV_CHECK_ARG(x || (n_partials >= 1 && n_partials <= BN_MAX_WGS), "sv_batchnorm_finalize_forward: x, or 1..%d partials (got %d)", BN_MAX_WGS, n_partials);
  BnArgs a{};
  a.x = x, a.gamma = gamma, a.beta = beta, a.running_mean = running_mean, a.running_var = running_var;
  a.save_mean = save_mean, a.save_invstd = save_invstd, a.n = n, a.C = channels, a.momentum = momentum, a.eps = eps;
  a.wgs = x ? bn_wgs(n, channels) : n_partials;
  a.num_batches_tracked = num_batches_tracked;
  bn_scratch(a, scratch);
  a.coef = coef;                                           // k_bn_finalize<false> writes coef[c] and coef[C + c] only
  if (x) hipLaunchKernelGGL(k_bn_reduce<false>, dim3(a.wgs), dim3(BN_THREADS), 0, sv_stream(stream), a);       // no producer-side partials: the statistics pass runs here
  hipLaunchKernelGGL(k_bn_finalize<false>, dim3(channels), dim3(BN_THREADS), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// y = [relu](x * scale + shift) with coef (2, C) = scale | shift: the elementwise pass of the forward on its own (the same kernel, the same expression)
extern "C" int sv_batchnorm_apply(const float* x, int64_t n, int channels, const float* coef, int relu, float* y, void* stream) {
  if (int rc = bn_common_check("sv_batchnorm_apply", n, channels)) return rc;
  SV_CHECK_ARG(x && coef && y && (uintptr_t)coef % 16 == 0, "sv_batchnorm_apply: null or misaligned pointer");
  BnArgs a{};
  a.x = x, a.out = y, a.n = n, a.C = channels, a.relu = relu, a.coef = const_cast<float*>(coef);
  hipLaunchKernelGGL(k_bn_apply_fwd, dim3(sv_grid_1d(n * (channels / 4), BN_THREADS)), dim3(BN_THREADS), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_batchnorm_relu_backward(const float* x, const float* dy, int64_t n, int channels, const float* gamma, const float* beta,
                                          const float* save_mean, const float* save_invstd, int relu, void* scratch, float* dx,
                                          float* dgamma, float* dbeta, void* stream) {
  if (int rc = bn_common_check("sv_batchnorm_relu_backward", n, channels)) return rc;
  SV_CHECK_ARG(x && dy && dx && dgamma && dbeta && save_mean && save_invstd && scratch, "sv_batchnorm_relu_backward: null pointer");
  BnArgs a{};
  a.x = x, a.dy = dy, a.out = dx, a.gamma = gamma, a.beta = beta, a.dgamma = dgamma, a.dbeta = dbeta;
  a.save_mean = const_cast<float*>(save_mean), a.save_invstd = const_cast<float*>(save_invstd);
  a.n = n, a.C = channels, a.relu = relu;
  a.wgs = bn_wgs(n, channels);
  bn_scratch(a, scratch);
  hipStream_t st = sv_stream(stream);
  hipLaunchKernelGGL(k_bn_reduce<true>, dim3(a.wgs), dim3(BN_THREADS), 0, st, a);
  bn_backward_tail(a, st);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// The same backward with the two per-channel sums' first pass already done: `scratch` holds n_partials per-workgroup partial sums
// (n_partials, 2, C) = {sum of masked dy, sum of masked dy * xhat} behind its 4 * C coefficient floats -- written there by the epilogue of the
// data-gradient convolution that produced dy (sv_sparse_conv_dgrad_planned_bn).  Two launches instead of three, x and dy read once.
extern "C" int sv_batchnorm_relu_backward_partial(const float* x, const float* dy, int64_t n, int channels, const float* gamma, const float* beta,
                                                  const float* save_mean, const float* save_invstd, int relu, void* scratch, int n_partials, float* dx,
                                                  float* dgamma, float* dbeta, void* stream) {
  if (int rc = bn_common_check("sv_batchnorm_relu_backward_partial", n, channels)) return rc;
  SV_CHECK_ARG(x && dy && dx && dgamma && dbeta && save_mean && save_invstd && scratch, "sv_batchnorm_relu_backward_partial: null pointer");
  SV_CHECK_ARG(n_partials >= 1 && n_partials <= BN_MAX_WGS, "sv_batchnorm_relu_backward_partial: 1..%d partials (got %d)", BN_MAX_WGS, n_partials);
  BnArgs a{};
  a.x = x, a.dy = dy, a.out = dx, a.gamma = gamma, a.beta = beta, a.dgamma = dgamma, a.dbeta = dbeta;
  a.save_mean = const_cast<float*>(save_mean), a.save_invstd = const_cast<float*>(save_invstd);
  a.n = n, a.C = channels, a.relu = relu;
  a.wgs = n_partials;
  bn_scratch(a, scratch);
  hipStream_t st = sv_stream(stream);
  bn_backward_tail(a, st);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// internal (norm.h): the per-channel combine of workgroup partials, for kernels that produce the partial sums in their own epilogue
// (set_abstraction_train.hip).  a.partial (wgs, 2, C), a.coef (4, C), a.n = number of rows the statistics run over.
void sv_bn_finalize_fwd(const BnArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_bn_finalize<false>, dim3(a.C), dim3(BN_THREADS), 0, st, a); }
void sv_bn_finalize_bwd(const BnArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_bn_finalize<true>, dim3(a.C), dim3(BN_THREADS), 0, st, a); }
