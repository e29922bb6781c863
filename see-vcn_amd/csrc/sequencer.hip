// Launch-list executor: one C-ABI call that enqueues a whole list of this library's operations on a stream.
//
// The trained side of a step (VoxelBackBone8x: 13 x [sparse conv -> BatchNorm -> ReLU], and the reverse chain of BatchNorm backward, weight
// gradient and data gradient) is ~65 launches whose arguments are all known before the first one runs (the rulebooks and plans were built
// ahead, the buffers are slices of one allocation).  Enqueued from Python one call at a time they cost the host 2.2 ms per step -- autograd
// node, output allocations and a ctypes call per layer and direction -- next to 4.0 ms of GPU time; the reference pays the same per-layer
// price inside spconv's Python autograd functions (spconv_backbone.py:128-180 is the caller).  Here the host writes the list as rows of 32
// int64 (pointers, sizes, flags; floats as the bits of a double) and hands it over once: seevcn_amd/spconv/chain.py builds the forward and the
// backward list of a conv -> norm -> ReLU chain and runs each with one sv_run_ops call inside ONE autograd node.
#include <string.h>

#include "common.h"

namespace {
inline double as_double(int64_t bits) {
  double d;
  memcpy(&d, &bits, sizeof d);
  return d;
}
template <typename T>
inline T* ptr_of(int64_t v) { return reinterpret_cast<T*>(static_cast<uintptr_t>(v)); }
}   // namespace

// Events for the two-stream form live for ONE call (created on the caller's current device, destroyed before the call returns: destroying an event
// that is still pending only defers its release): no state shared between devices or host threads.
struct SeqEvents {
  hipEvent_t fork = nullptr, join = nullptr;
  int create() {
    SV_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    SV_HIP(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    return SV_OK;
  }
  ~SeqEvents() {
    if (fork) (void)hipEventDestroy(fork);
    if (join) (void)hipEventDestroy(join);
  }
};

static int run_ops(const int64_t* ops, int n_ops, void* stream, void* side_stream, float* ms = nullptr);

extern "C" int sv_run_ops(const int64_t* ops, int n_ops, void* stream) { return run_ops(ops, n_ops, stream, nullptr); }

// Measurement form: the same list, with a HIP event recorded on `stream` in front of every operation and behind the last; the call then WAITS for the
// stream and writes the elapsed time of every operation to ms[0 .. n_ops) (an operation of several launches -- BatchNorm backward -- as one span; the
// deferred slab reduction at the end of the list belongs to no operation).  bench.py times the conv launches of the step's own launch lists with it.
extern "C" int sv_run_ops_timed(const int64_t* ops, int n_ops, void* stream, float* ms) {
  SV_CHECK_ARG(ms || n_ops == 0, "sv_run_ops_timed: null output");
  return run_ops(ops, n_ops, stream, nullptr, ms);
}

// The same list with its weight gradients (SV_OP_WGRAD, SV_OP_WGRAD_DEFERRED and the deferred reduction) on `side_stream`: each goes behind an event
// recorded on `stream` after the operations in front of it, the other operations do not wait for it, and `stream` waits for `side_stream` once, at the
// end -- when the call returns everything is ordered on `stream` as after sv_run_ops.  A weight gradient only feeds the optimiser: the backward chain
// (BatchNorm backward -> data gradient -> next BatchNorm backward) need not stop for it, and its matrix-core work fills the bandwidth-bound BatchNorm
// launches and the tails of the data-gradient launches (trained side of the benchmarked step alone: 3.16 -> 3.05 ms).  Same kernels, same results.
extern "C" int sv_run_ops_two_streams(const int64_t* ops, int n_ops, void* stream, void* side_stream) {
  SV_CHECK_ARG(side_stream && side_stream != stream, "sv_run_ops_two_streams: needs a second stream");
  return run_ops(ops, n_ops, stream, side_stream);
}

static int run_ops(const int64_t* ops, int n_ops, void* stream, void* side_stream, float* ms) {
  SV_CHECK_ARG(n_ops >= 0 && (ops || n_ops == 0), "sv_run_ops: null list");
  struct Stamps {                    // measurement form only
    hipEvent_t* ev = nullptr;
    int n = 0;
    ~Stamps() {
      for (int q = 0; q < n; ++q) (void)hipEventDestroy(ev[q]);
      delete[] ev;
    }
  } stamps;
  if (ms) {
    stamps.ev = new hipEvent_t[n_ops + 1];
    for (; stamps.n <= n_ops; ++stamps.n) SV_HIP(hipEventCreate(&stamps.ev[stamps.n]));
    SV_HIP(hipEventRecord(stamps.ev[0], sv_stream(stream)));
  }
  hipStream_t st_main = sv_stream(stream), st_side = side_stream ? sv_stream(side_stream) : nullptr;
  bool side_used = false;
  SeqEvents ev;
  if (st_side) {
    if (int rc = ev.create()) return rc;
  }
  // side stream behind everything enqueued on the main stream so far; a failure here is a failure of the operation (the list stops, the join below still runs)
  auto fork_side = [&]() -> int {
    SV_HIP(hipEventRecord(ev.fork, st_main));
    side_used = true;                                     // from here on the side stream may hold work of this call
    SV_HIP(hipStreamWaitEvent(st_side, ev.fork, 0));
    return SV_OK;
  };
  void* const wstream = side_stream ? side_stream : stream;      // where weight gradients go
  constexpr int MAX_DEFERRED = 64;
  int64_t deferred[MAX_DEFERRED * 10];       // stage-2 jobs of the SV_OP_WGRAD_DEFERRED operations of this list
  int n_deferred = 0, fail = SV_OK;
  for (int k = 0; k < n_ops; ++k) {
    const int64_t* o = ops + (size_t)k * SV_OP_WORDS;
    const int64_t* i = o + 1;       // 8 small integers
    const int64_t* n = o + 9;       // 4 sizes / strides
    const int64_t* f = o + 13;      // 4 doubles (bit patterns)
    const int64_t* p = o + 17;      // 15 pointers
    int rc = SV_OK;
    switch ((int)o[0]) {
      case SV_OP_CONV_PLANNED:
        if (p[12]) sv_conv_next_input_norm(ptr_of<const float>(p[12]), (int)i[6]);        // X = raw conv output of the layer below, its BatchNorm applied on load
        rc = sv_sparse_conv_gather_gemm_planned(ptr_of<const float>(p[0]), n[0], ptr_of<const int32_t>(p[1]), ptr_of<const int32_t>(p[2]),
                                                ptr_of<const int32_t>(p[3]), ptr_of<const int32_t>(p[4]), (int)i[0], ptr_of<const float>(p[5]),
                                                ptr_of<float>(p[6]), n[1], (int)i[1], (int)i[2], (int)i[3], ptr_of<const float>(p[7]),
                                                ptr_of<const float>(p[8]), ptr_of<const float>(p[9]), ptr_of<const float>(p[10]), (int)i[4], (int)i[5],
                                                ptr_of<float>(p[11]), stream);
        break;
      case SV_OP_CONV_PLAIN:
        rc = sv_sparse_conv_gather_gemm(ptr_of<const float>(p[0]), n[0], ptr_of<const int32_t>(p[1]), ptr_of<const float>(p[2]), ptr_of<float>(p[3]), n[1],
                                        (int)i[0], (int)i[1], (int)i[2], ptr_of<const float>(p[4]), ptr_of<const float>(p[5]), ptr_of<const float>(p[6]),
                                        ptr_of<const float>(p[7]), (int)i[3], stream);
        break;
      case SV_OP_BN_FWD:
        if (i[3] > 0)
          rc = sv_batchnorm_relu_forward_partial(ptr_of<const float>(p[0]), n[0], (int)i[0], ptr_of<const float>(p[1]), ptr_of<const float>(p[2]),
                                                 ptr_of<float>(p[3]), ptr_of<float>(p[4]), (float)as_double(f[0]), (float)as_double(f[1]), (int)i[2],
                                                 ptr_of<void>(p[5]), (int)i[3], ptr_of<float>(p[6]), ptr_of<float>(p[7]), ptr_of<float>(p[8]),
                                                 ptr_of<int64_t>(p[9]), stream);
        else
          rc = sv_batchnorm_relu_forward(ptr_of<const float>(p[0]), n[0], (int)i[0], ptr_of<const float>(p[1]), ptr_of<const float>(p[2]), ptr_of<float>(p[3]),
                                         ptr_of<float>(p[4]), (float)as_double(f[0]), (float)as_double(f[1]), (int)i[1], (int)i[2], ptr_of<void>(p[5]),
                                         ptr_of<float>(p[6]), ptr_of<float>(p[7]), ptr_of<float>(p[8]), ptr_of<int64_t>(p[9]), stream);
        break;
      case SV_OP_DGRAD_PLANNED_BN:
        rc = sv_sparse_conv_dgrad_planned_bn(ptr_of<const float>(p[0]), n[0], ptr_of<const int32_t>(p[1]), ptr_of<const int32_t>(p[2]),
                                             ptr_of<const int32_t>(p[3]), ptr_of<const int32_t>(p[4]), (int)i[0], ptr_of<const float>(p[5]),
                                             ptr_of<float>(p[6]), n[1], (int)i[1], (int)i[2], (int)i[3], (int)i[4], ptr_of<const float>(p[7]),
                                             ptr_of<const float>(p[8]), ptr_of<const float>(p[9]), ptr_of<const float>(p[10]), ptr_of<const float>(p[11]),
                                             (int)i[5], ptr_of<float>(p[12]), stream);
        break;
      case SV_OP_BN_BWD:
        if (i[2] > 0)
          rc = sv_batchnorm_relu_backward_partial(ptr_of<const float>(p[0]), ptr_of<const float>(p[1]), n[0], (int)i[0], ptr_of<const float>(p[2]),
                                                  ptr_of<const float>(p[3]), ptr_of<const float>(p[4]), ptr_of<const float>(p[5]), (int)i[1],
                                                  ptr_of<void>(p[6]), (int)i[2], ptr_of<float>(p[7]), ptr_of<float>(p[8]), ptr_of<float>(p[9]), stream);
        else
        rc = sv_batchnorm_relu_backward(ptr_of<const float>(p[0]), ptr_of<const float>(p[1]), n[0], (int)i[0], ptr_of<const float>(p[2]),
                                        ptr_of<const float>(p[3]), ptr_of<const float>(p[4]), ptr_of<const float>(p[5]), (int)i[1], ptr_of<void>(p[6]),
                                        ptr_of<float>(p[7]), ptr_of<float>(p[8]), ptr_of<float>(p[9]), stream);
        break;
      case SV_OP_BN_FINALIZE: {
        // MEASUREMENT ONLY (results are stale after the first calls): SEEVCN_DEBUG_SKIP_FINALIZE=n skips the launch from the n-th call on -- what the step would
        // gain if the statistics' combine cost the chain nothing
        static const int skip_after = getenv("SEEVCN_DEBUG_SKIP_FINALIZE") ? atoi(getenv("SEEVCN_DEBUG_SKIP_FINALIZE")) : 0;
        static int calls = 0;
        if (skip_after > 0 && ++calls > skip_after && !p[9]) break;
        rc = sv_batchnorm_finalize_forward(ptr_of<const float>(p[9]), n[0], (int)i[0], ptr_of<const float>(p[0]), ptr_of<const float>(p[1]), ptr_of<float>(p[2]), ptr_of<float>(p[3]),
                                           (float)as_double(f[0]), (float)as_double(f[1]), ptr_of<void>(p[4]), (int)i[1], ptr_of<float>(p[5]), ptr_of<float>(p[6]),
                                           ptr_of<float>(p[7]), ptr_of<int64_t>(p[8]), stream);
        break;
      }
      case SV_OP_BN_APPLY:
        rc = sv_batchnorm_apply(ptr_of<const float>(p[0]), n[0], (int)i[0], ptr_of<const float>(p[1]), (int)i[1], ptr_of<float>(p[2]), stream);
        break;
      case SV_OP_WGRAD:
        if (st_side && (rc = fork_side()) != SV_OK) break;
        if (p[6]) sv_conv_next_input_norm(ptr_of<const float>(p[6]), (int)i[4]);
        if (p[5])
          rc = sv_sparse_conv_wgrad_planned(ptr_of<const float>(p[0]), i[3], ptr_of<const int32_t>(p[1]), ptr_of<const float>(p[2]), ptr_of<float>(p[3]), n[0],
                                            (int)i[0], (int)i[1], (int)i[2], n[1], n[2], n[3], ptr_of<const void>(p[5]), ptr_of<void>(p[4]), wstream);
        else
        rc = sv_sparse_conv_wgrad_strided(ptr_of<const float>(p[0]), i[3], ptr_of<const int32_t>(p[1]), ptr_of<const float>(p[2]), ptr_of<float>(p[3]), n[0],
                                          (int)i[0], (int)i[1], (int)i[2], n[1], n[2], n[3], ptr_of<void>(p[4]), wstream);
        break;
      case SV_OP_WGRAD_DEFERRED:
        if (st_side && (rc = fork_side()) != SV_OK) break;
        if (n_deferred == MAX_DEFERRED) {     // more than a list's worth: sum what is pending, go on
          rc = sv_sparse_conv_wgrad_reduce_batch(deferred, n_deferred, wstream);
          n_deferred = 0;
          if (rc != SV_OK) break;
        }
        if (p[6]) sv_conv_next_input_norm(ptr_of<const float>(p[6]), (int)i[4]);
        if (p[5])
          rc = sv_sparse_conv_wgrad_planned_stage1(ptr_of<const float>(p[0]), i[3], ptr_of<const int32_t>(p[1]), ptr_of<const float>(p[2]), ptr_of<float>(p[3]), n[0],
                                                   (int)i[0], (int)i[1], (int)i[2], n[1], n[2], n[3], ptr_of<const void>(p[5]), ptr_of<void>(p[4]),
                                                   deferred + 10 * n_deferred, wstream);
        else
        rc = sv_sparse_conv_wgrad_stage1(ptr_of<const float>(p[0]), i[3], ptr_of<const int32_t>(p[1]), ptr_of<const float>(p[2]), ptr_of<float>(p[3]), n[0],
                                         (int)i[0], (int)i[1], (int)i[2], n[1], n[2], n[3], ptr_of<void>(p[4]), deferred + 10 * n_deferred, wstream);
        ++n_deferred;
        break;
      default:
        sv_set_error("sv_run_ops: unknown operation %lld at position %d", (long long)o[0], k);
        rc = SV_ERR_ARG;
        break;
    }
    if (rc != SV_OK) {              // sv_last_error() names the failing entry point; `k` operations were enqueued
      fail = rc;
      break;
    }
    if (ms) SV_HIP(hipEventRecord(stamps.ev[k + 1], st_main));
  }
  int rc = fail;
  sv_conv_next_input_norm(nullptr, 0);                    // an operation that failed before it consumed its input transform must not leave it to a later call
  if (rc == SV_OK && n_deferred > 0) rc = sv_sparse_conv_wgrad_reduce_batch(deferred, n_deferred, wstream);     // every deferred weight gradient: one launch at the end of the list
  if (ms && rc == SV_OK) {
    SV_HIP(hipStreamSynchronize(st_main));
    for (int k = 0; k < n_ops; ++k) SV_HIP(hipEventElapsedTime(&ms[k], stamps.ev[k], stamps.ev[k + 1]));
  }
  if (side_used) {                  // whatever happened above: the caller's stream comes back ordered behind the side stream
    hipError_t e = hipEventRecord(ev.join, st_side);
    if (e == hipSuccess) e = hipStreamWaitEvent(st_main, ev.join, 0);
    if (e != hipSuccess) {          // cannot order the streams by event: wait for the side stream on the host so that no buffer of this call is still in use on return
      (void)hipStreamSynchronize(st_side);
      if (rc == SV_OK) {
        sv_set_error("sv_run_ops_two_streams: joining the side stream failed: %s", hipGetErrorString(e));
        rc = SV_ERR_HIP;
      }
    }
  }
  return rc;
}
