"""GPU versions of the reference's CPU post-processing of VCN outputs, same names and return types
(see/surface_completion/models/vcn/utils/sampling.py:8-110).  The `*_device` functions keep everything on the GPU
(torch tensors in and out) so VCN.inference can chain them without a host round trip."""
import numpy as np
import torch

from ... import _lib


def _as_device_f32(x, device=None):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if device is not None and x.device.type != 'cuda':
        x = x.to(device)
    return x.detach().float().contiguous()


def get_partial_mesh_batch_device(batch_partial, batch_complete, k=20, surface_pts=1024):
    """(B,n,3), (B,m,3) CUDA tensors -> surface (B,surface_pts,3) float32 tensor, n_selected (B) int32 tensor."""
    lib = _lib.load()
    _lib.require_cuda(batch_partial, batch_complete)
    p, c = _as_device_f32(batch_partial), _as_device_f32(batch_complete)
    assert p.dim() == 3 and c.dim() == 3 and p.shape[0] == c.shape[0] and p.shape[2] == 3 and c.shape[2] == 3
    B = p.shape[0]
    out = torch.empty((B, surface_pts, 3), dtype=torch.float32, device=p.device)
    nsel = torch.empty((B,), dtype=torch.int32, device=p.device)
    scratch = _lib.workspace.scratch("surface_select", lib.sv_vcn_surface_select_scratch_bytes(B), p.device)
    _lib.check(lib.sv_vcn_surface_select(_lib.ptr(p), _lib.ptr(c), B, p.shape[1], c.shape[1], int(k), int(surface_pts), _lib.ptr(scratch), _lib.ptr(out),
                                         _lib.ptr(nsel), _lib.stream()), "sv_vcn_surface_select")
    return out, nsel


def partial_with_KDTree(partial_pc, complete_pc, k, surface_pts=1024):
    assert len(partial_pc.shape) == 2, f'partial_pc shape is {partial_pc.shape}, must have shape (1024,3)'
    dev = complete_pc.device if isinstance(complete_pc, torch.Tensor) and complete_pc.is_cuda else partial_pc.device
    out, _ = get_partial_mesh_batch_device(_as_device_f32(partial_pc, dev)[None], _as_device_f32(complete_pc, dev)[None], k, surface_pts)
    return out[0].cpu().numpy()


def get_partial_mesh_batch(batch_partial, batch_complete, k=20, surface_pts=1024):
    """Reference signature: returns a numpy (B,surface_pts,3) float32 array."""
    return get_partial_mesh_batch_device(batch_partial, batch_complete, k, surface_pts)[0].cpu().numpy()


def _no_cluster(smallest):
    if smallest == 0:
        raise ValueError("attempt to get argmax of an empty sequence")


def get_largest_cluster_batch_device(pc, eps=0.4, min_points=1, total_pts=1024, defer_check=False, period=None):
    """(B,n,3) CUDA tensor -> (B,total_pts,3) float32 tensor, cluster sizes (B) int32.  Raises ValueError (like the reference's
    np.argmax over an empty bincount) if some object has no cluster at all -- at once, or with defer_check at the caller's next blocking
    read on this stream (_lib.defer_check: the pipeline's voxel count follows within the same step).  period (B) int32 CUDA tensor, optional:
    object b's cloud repeats its first period[b] rows (get_partial_mesh_batch_device's second return value) -- a hint that is verified on the
    device and only shortens the pair tests (same output)."""
    lib = _lib.load()
    _lib.require_cuda(pc)
    x = _as_device_f32(pc)
    assert x.dim() == 3 and x.shape[2] == 3
    B = x.shape[0]
    out = torch.zeros((B, total_pts, 3), dtype=torch.float32, device=x.device)
    cnt = torch.empty((B,), dtype=torch.int32, device=x.device)
    if period is not None:
        assert period.dtype == torch.int32 and period.shape == (B,) and period.is_cuda
        _lib.check(lib.sv_vcn_largest_cluster_periodic(_lib.ptr(x), B, x.shape[1], _lib.ptr(period.contiguous()), float(eps), int(min_points), int(total_pts),
                                                       _lib.ptr(out), _lib.ptr(cnt), _lib.stream()), "sv_vcn_largest_cluster_periodic")
    else:
        _lib.check(lib.sv_vcn_largest_cluster(_lib.ptr(x), B, x.shape[1], float(eps), int(min_points), int(total_pts), _lib.ptr(out), _lib.ptr(cnt),
                                              _lib.stream()), "sv_vcn_largest_cluster")
    if B:
        if defer_check:
            _lib.defer_check(cnt.min(), _no_cluster)
        else:
            _no_cluster(_lib.host_int(cnt.min()))
    return out, cnt


def get_largest_cluster(pc, eps=0.4, min_points=1, istensor=False, total_pts=1024, device='cuda'):
    return get_largest_cluster_batch_device(_as_device_f32(pc, device)[None], eps, min_points, total_pts)[0][0].cpu().numpy().astype(np.float64)


def get_largest_cluster_batch(pc, eps=0.4, min_points=1, total_pts=1024, device='cuda'):
    """Reference signature: numpy (B,N,3) in, numpy float64 (B,total_pts,3) out."""
    return get_largest_cluster_batch_device(_as_device_f32(pc, device), eps, min_points, total_pts)[0].cpu().numpy().astype(np.float64)
