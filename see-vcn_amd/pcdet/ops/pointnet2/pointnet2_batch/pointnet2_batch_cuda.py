"""`pointnet2_batch_cuda` with the reference's wrapper names and argument order
(detector3d/pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:12-27) over libseevcn_hip.so."""
from ..... import _lib


def _call(name, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args, _lib.stream()), name)
    return 1


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    _lib.require_cuda(new_xyz, xyz, idx)
    return _call("sv_ball_query_batch", int(b), int(n), int(m), float(radius), int(nsample), _lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(idx))


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _lib.require_cuda(points, idx, out)
    return _call("sv_group_points_batch", int(b), int(c), int(n), int(npoints), int(nsample), _lib.ptr(points), _lib.ptr(idx), _lib.ptr(out))


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _lib.require_cuda(grad_out, idx, grad_points)
    return _call("sv_group_points_grad_batch", int(b), int(c), int(n), int(npoints), int(nsample), _lib.ptr(grad_out), _lib.ptr(idx),
                 _lib.ptr(grad_points))


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _lib.require_cuda(points, idx, out)
    return _call("sv_gather_points_batch", int(b), int(c), int(n), int(npoints), _lib.ptr(points), _lib.ptr(idx), _lib.ptr(out))


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _lib.require_cuda(grad_out, idx, grad_points)
    return _call("sv_gather_points_grad_batch", int(b), int(c), int(n), int(npoints), _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(grad_points))


def farthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    _lib.require_cuda(points, idx)
    from ..pointnet2_stack.pointnet2_stack_cuda import fps_bucketed
    if fps_bucketed(points, None, None, b, n, n, m, idx):
        return 1
    return _call("sv_farthest_point_sampling", _lib.ptr(points), int(b), int(n), int(m), _lib.ptr(temp), _lib.ptr(idx))


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _lib.require_cuda(unknown, known, dist2, idx)
    return _call("sv_three_nn_batch", int(b), int(n), int(m), _lib.ptr(unknown), _lib.ptr(known), _lib.ptr(dist2), _lib.ptr(idx))


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _lib.require_cuda(points, idx, weight, out)
    return _call("sv_three_interpolate_batch", int(b), int(c), int(m), int(n), _lib.ptr(points), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(out))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _lib.require_cuda(grad_out, idx, weight, grad_points)
    return _call("sv_three_interpolate_grad_batch", int(b), int(c), int(n), int(m), _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(weight),
                 _lib.ptr(grad_points))
