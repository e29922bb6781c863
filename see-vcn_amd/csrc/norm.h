// Shared by norm.hip and set_abstraction_train.hip: the argument block of the fused BatchNorm kernels and the launchers of the
// per-channel fp64 fixed-order combine (k_bn_finalize) for callers that make the workgroup partial sums themselves.
#pragma once
#include "common.h"

#define BN_THREADS 256
#define BN_MAX_C 512
#define BN_MAX_WGS 1024

struct BnArgs {
  const float* x;       // (N, C) BN input
  const float* dy;      // (N, C) gradient wrt the output (backward only)
  float* out;           // y (forward) or dx (backward)
  const float* gamma;   // may be null (1)
  const float* beta;    // may be null (0)
  float* running_mean;  // may be null
  float* running_var;
  float* save_mean;     // (C)
  float* save_invstd;   // (C)
  float* dgamma;        // (C)
  float* dbeta;
  float* partial;       // scratch: (wgs, 2, C)
  float* coef;          // scratch: (4, C) per-channel constants of the elementwise pass
  int64_t n;
  int C, relu, wgs;
  float momentum, eps;
  int64_t* num_batches_tracked;   // nn.BatchNorm1d's counter, incremented by the training forward (may be null)
};

// y = x * scale + shift, scale = invstd * gamma, shift = beta - mean * scale.  Every place that needs the ReLU branch of the forward (the backward
// passes recompute it from x) must evaluate EXACTLY this expression -- same operations, same roundings -- or an activation within an ulp of zero
// is "off" in the forward and "on" in the backward.  Hence explicit fused multiply-adds (no compiler contraction choices) in one place.
__device__ __forceinline__ float bn_shift(float mean, float scale, float beta) { return __fmaf_rn(-mean, scale, beta); }
__device__ __forceinline__ float bn_act(float x, float scale, float shift) { return __fmaf_rn(x, scale, shift); }

void sv_bn_finalize_fwd(const BnArgs& a, hipStream_t st);   // partial (wgs,2,C) -> save_mean, save_invstd, running stats, coef = {scale, shift}
void sv_bn_finalize_bwd(const BnArgs& a, hipStream_t st);   // partial -> dgamma, dbeta, coef = {gamma*invstd, mean(dy), mean(dy*xhat), mean}
