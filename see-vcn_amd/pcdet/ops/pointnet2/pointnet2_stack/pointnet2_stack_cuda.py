"""Same entry-point names and argument order as the reference's compiled extension `pointnet2_stack_cuda`
(detector3d/pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:12-31), bound to libseevcn_hip.so.
Outputs are caller-allocated torch tensors, every function returns 1 like the reference wrappers."""
import os

import torch

from ..... import _lib


def _version(t):
    try:
        return t._version
    except RuntimeError:                                  # inference tensors keep no version counter: nothing is cached on them
        return None


def _starts(cnt):
    """(first row of every scene, rows per scene) as int32 device tensors.  A step asks for the same few count tensors again and again (every
    scale of every source: 25 cumsum + 25 sub launches per PV-RCNN step), so the pair is kept on the count tensor OBJECT, keyed by its
    version counter (an in-place change invalidates it)."""
    ver = _version(cnt)
    hit = getattr(cnt, "_sv_starts", None) if ver is not None else None
    if hit is not None and hit[0] == ver:
        return hit[1], hit[2]
    c32 = cnt.to(torch.int32).contiguous()
    st = (torch.cumsum(c32, 0, dtype=torch.int32) - c32).contiguous()
    if ver is not None:
        try:
            cnt._sv_starts = (ver, st, c32)
        except Exception:                                 # a tensor subclass without instance attributes: no cache
            pass
    return st, c32


BALL_HASH_MIN_POINTS = int(os.environ.get("SEEVCN_BALL_HASH_MIN", "2048"))   # support sets below this (or nsample > 64) are scanned like the reference does


def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
    lib = _lib.load()
    _lib.require_cuda(new_xyz, xyz, idx)
    qs, qc = _starts(new_xyz_batch_cnt)
    ps, pc = _starts(xyz_batch_cnt)
    n = int(xyz.shape[0])
    if n >= BALL_HASH_MIN_POINTS and nsample <= 64 and radius > 0:
        # the same idx, element for element, from a cell hash of the support points (sv_ball_query_stack_hashed) instead of the O(M N) scan
        scratch = _lib.workspace.scratch("ball_hash", lib.sv_ball_query_hash_scratch_bytes(n), xyz.device)
        rc = lib.sv_ball_query_stack_hashed(int(B), int(M), n, float(radius), int(nsample), _lib.ptr(new_xyz), _lib.ptr(qs), _lib.ptr(qc), _lib.ptr(xyz),
                                            _lib.ptr(ps), _lib.ptr(pc), _lib.ptr(scratch), _lib.ptr(idx), _lib.stream())
        _lib.check(rc, "sv_ball_query_stack_hashed")
        return 1
    max_q = int(M)  # upper bound of queries per scene without a host sync
    rc = lib.sv_ball_query_stack(int(B), int(M), max_q, float(radius), int(nsample), _lib.ptr(new_xyz), _lib.ptr(qs), _lib.ptr(qc),
                                 _lib.ptr(xyz), _lib.ptr(ps), _lib.ptr(pc), _lib.ptr(idx), _lib.stream())
    _lib.check(rc, "sv_ball_query_stack")
    return 1


def _row_start(idx_batch_cnt, features_batch_cnt, M):
    """first feature row of the scene each query belongs to (M,) int32"""
    vf, vi = _version(features_batch_cnt), _version(idx_batch_cnt)
    cacheable = vf is not None and vi is not None
    cache = getattr(idx_batch_cnt, "_sv_row_start", None) if cacheable else None
    key = (id(features_batch_cnt), vf, vi, int(M))
    if cache is not None and cache[0] == key and cache[1] is features_batch_cnt:
        return cache[2]
    fs, _ = _starts(features_batch_cnt)
    rs = torch.repeat_interleave(fs, idx_batch_cnt.long(), output_size=int(M)).contiguous()
    if cacheable:
        try:
            idx_batch_cnt._sv_row_start = (key, features_batch_cnt, rs)  # the last support set asked for with these queries (scales come in a row)
        except Exception:
            pass
    return rs


def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
    lib = _lib.load()
    _lib.require_cuda(features, idx, out)
    rs = _row_start(idx_batch_cnt, features_batch_cnt, M)
    rc = lib.sv_group_points_stack(int(M), int(C), int(nsample), _lib.ptr(features), _lib.ptr(idx), _lib.ptr(rs), _lib.ptr(out), _lib.stream())
    _lib.check(rc, "sv_group_points_stack")
    return 1


def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
    lib = _lib.load()
    _lib.require_cuda(grad_out, idx, grad_features)
    rs = _row_start(idx_batch_cnt, features_batch_cnt, M)
    rc = lib.sv_group_points_grad_stack(int(M), int(C), int(N), int(nsample), _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(rs),
                                        _lib.ptr(grad_features), _lib.stream())
    _lib.check(rc, "sv_group_points_grad_stack")
    return 1


def fps_bucketed(points, starts, cnt, batch, fixed_n, max_n, m, idx):
    """One workgroup per scene on spatial buckets (csrc/fps_bucket.hip) where it applies -> True; False: the caller takes the exhaustive kernels.
    Same indices either way."""
    lib = _lib.load()
    if not lib.sv_fps_bucket_applies(int(batch), int(max_n), int(m)):
        return False
    if cnt is not None:
        # the kernels size their scene slabs from the HOST's max_n: a scene longer than that (a stale points-per-scene bound) would be cut short without a
        # word -- the exhaustive kernels had their error word for this.  Parked: raised at the stream's next host read, no read of its own.
        def _check(v, max_n=int(max_n)):
            if v > max_n:
                raise _lib.SeevcnHipError(f"farthest_point_sampling: a scene has {v} points, the caller's max_n is {max_n}")
        _lib.defer_check(cnt.max(), _check)
    scratch = torch.empty((int(lib.sv_fps_bucket_scratch_bytes(int(batch), int(max_n))),), dtype=torch.uint8, device=points.device)
    _lib.check(lib.sv_farthest_point_sampling_bucketed(_lib.ptr(points), _lib.ptr(starts), _lib.ptr(cnt), int(batch), int(fixed_n), int(max_n), int(m),
                                                       _lib.ptr(scratch), _lib.ptr(idx), _lib.stream()), "sv_farthest_point_sampling_bucketed")
    return True


def farthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    lib = _lib.load()
    _lib.require_cuda(points, idx)
    if fps_bucketed(points, None, None, b, n, n, m, idx):
        return 1
    rc = lib.sv_farthest_point_sampling(_lib.ptr(points), int(b), int(n), int(m), _lib.ptr(temp), _lib.ptr(idx), _lib.stream())
    _lib.check(rc, "sv_farthest_point_sampling")
    return 1


def stack_farthest_point_sampling(points, xyz_batch_cnt, npoint, max_n=None, bucketed=True):
    """All ragged scenes in one launch -> (batch, npoint) int32 GLOBAL row indices (seevcn extension; replaces the
    per-scene loop of voxel_set_abstraction.py:250-256).  bucketed=False: the exhaustive kernels only (tests, A/B runs)."""
    lib = _lib.load()
    _lib.require_cuda(points)
    starts, cnt = _starts(xyz_batch_cnt)
    batch = cnt.shape[0]
    if max_n is None:
        max_n = int(cnt.max().item())
    idx = torch.empty((batch, npoint), dtype=torch.int32, device=points.device)
    if bucketed and fps_bucketed(points, starts, cnt, batch, 0, max_n, npoint, idx):
        return idx
    temp = torch.empty((points.shape[0],), dtype=torch.float32, device=points.device) if max_n > 24576 else None
    # several workgroups per scene where that applies (sv_stack_farthest_point_sampling_multi decides; same indices either way)
    nbytes = int(lib.sv_fps_multi_scratch_bytes(batch))
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=points.device)
    rc = lib.sv_stack_farthest_point_sampling_multi(_lib.ptr(points), _lib.ptr(starts), _lib.ptr(cnt), batch, int(max_n), int(npoint),
                                                    _lib.ptr(temp), _lib.ptr(scratch), _lib.ptr(idx), _lib.stream())
    _lib.check(rc, "sv_stack_farthest_point_sampling_multi")
    return idx


class FpsHandle:
    """A stacked farthest point sampling in flight (stack_farthest_point_sampling_async).  result() returns the (batch, npoint) int32 GLOBAL row
    indices; it reads the kernel's error word on the stream the sampling ran on (so only that stream is waited for) and samples once more with
    write-through records if a partner workgroup did not arrive."""

    def __init__(self, idx, err, stream, rerun):
        self.idx, self.err, self.stream, self._rerun = idx, err, stream, rerun
        self.done = torch.cuda.Event()         # behind the sampling on its stream: what result() orders the caller's stream after when there is no error word to read
        self.done.record(stream)

    def result(self):
        if self.err is None:                   # sampled by the one-workgroup bucket kernel: nothing that could have gone missing
            torch.cuda.current_stream().wait_event(self.done)      # no host wait: the caller's stream goes behind the sampling stream
            return self.idx
        with torch.cuda.stream(self.stream):
            bad = int(self.err.item())
            if bad:
                self._rerun()
                bad = int(self.err.item())
        if bad:
            raise _lib.SeevcnHipError("farthest_point_sampling: a partner workgroup never arrived (GPU shared with another process?); "
                                      "SEEVCN_FPS_MULTI=0 selects one workgroup per scene")
        return self.idx


def stack_farthest_point_sampling_async(points, xyz_batch_cnt, npoint, max_n):
    """stack_farthest_point_sampling without the read-back: enqueues on the current stream and returns an FpsHandle (seevcn extension, used to
    sample the keypoints on a side stream while the backbone runs).  max_n (largest scene) must come from the host."""
    lib = _lib.load()
    _lib.require_cuda(points)
    starts, cnt = _starts(xyz_batch_cnt)
    batch = cnt.shape[0]
    idx = torch.empty((batch, npoint), dtype=torch.int32, device=points.device)
    if fps_bucketed(points, starts, cnt, batch, 0, max_n, npoint, idx):
        return FpsHandle(idx, None, torch.cuda.current_stream(), None)
    temp = torch.empty((points.shape[0],), dtype=torch.float32, device=points.device) if max_n > 24576 else None
    nbytes = int(lib.sv_fps_multi_scratch_bytes(batch))
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=points.device)
    off = int(lib.sv_fps_multi_error_offset(batch))
    err = scratch[off:off + 4].view(torch.int32)
    stream = torch.cuda.current_stream()

    def launch(write_through):
        _lib.check(lib.sv_stack_farthest_point_sampling_multi_async(_lib.ptr(points), _lib.ptr(starts), _lib.ptr(cnt), batch, int(max_n), int(npoint),
                                                                    _lib.ptr(temp), _lib.ptr(scratch), _lib.ptr(idx), int(write_through), _lib.stream()),
                   "sv_stack_farthest_point_sampling_multi_async")

    launch(0)
    return FpsHandle(idx, err, stream, lambda: launch(1))


def _outside_hot_path(name):
    def stub(*args, **kwargs):
        raise NotImplementedError(f"pointnet2_stack_cuda.{name} is outside the SEE-VCN hot path (SURVEY.md 2: PV-RCNN++ / PartA2 / PointRCNN-style "
                                  f"heads) and not built; see INTEGRATION.md 'Unsupported surface'")
    stub.__name__ = name
    return stub


# names the reference's extension exports (src/pointnet2_api.cpp:12-31) that no SEE-VCN configuration reaches
three_nn_wrapper = _outside_hot_path("three_nn_wrapper")
three_interpolate_wrapper = _outside_hot_path("three_interpolate_wrapper")
three_interpolate_grad_wrapper = _outside_hot_path("three_interpolate_grad_wrapper")
voxel_query_wrapper = _outside_hot_path("voxel_query_wrapper")
vector_pool_wrapper = _outside_hot_path("vector_pool_wrapper")
vector_pool_grad_wrapper = _outside_hot_path("vector_pool_grad_wrapper")
query_stacked_local_neighbor_idxs_wrapper_stack = _outside_hot_path("query_stacked_local_neighbor_idxs_wrapper_stack")
query_three_nn_by_stacked_local_idxs_wrapper_stack = _outside_hot_path("query_three_nn_by_stacked_local_idxs_wrapper_stack")
