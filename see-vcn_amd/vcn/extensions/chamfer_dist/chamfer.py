"""Entry points of the reference's compiled extension `chamfer` (see/surface_completion/models/vcn/extensions/chamfer_dist/
chamfer_cuda.cpp:36-39; kernels chamfer.cu:15-201), same names, argument order and returned lists, bound to libseevcn_hip.so:

  forward(xyz1 (B,n,3), xyz2 (B,m,3)) -> [dist1 (B,n), dist2 (B,m), idx1 (B,n) int32, idx2 (B,m) int32]        chamfer.cu:96-121
  backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2) -> [grad_xyz1 (B,n,3), grad_xyz2 (B,m,3)]          chamfer.cu:168-201

`import chamfer` in the reference's chamfer_dist/__init__.py:10 binds to this module unchanged."""
import torch

from .... import _lib


def forward(xyz1, xyz2):
    lib = _lib.load()
    _lib.require_cuda(xyz1, xyz2)
    xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
    assert xyz1.dim() == 3 and xyz2.dim() == 3 and xyz1.shape[2] == 3 and xyz2.shape[2] == 3 and xyz1.shape[0] == xyz2.shape[0]
    B, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
    dev = xyz1.device
    dist1, dist2 = torch.empty((B, n), dtype=torch.float32, device=dev), torch.empty((B, m), dtype=torch.float32, device=dev)
    idx1, idx2 = torch.empty((B, n), dtype=torch.int32, device=dev), torch.empty((B, m), dtype=torch.int32, device=dev)
    _lib.check(lib.sv_chamfer_forward(_lib.ptr(xyz1), _lib.ptr(xyz2), B, n, m, _lib.ptr(dist1), _lib.ptr(dist2), _lib.ptr(idx1), _lib.ptr(idx2),
                                      _lib.stream()), "sv_chamfer_forward")
    return [dist1, dist2, idx1, idx2]


def backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2):
    lib = _lib.load()
    _lib.require_cuda(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2)
    xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
    B, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
    g1, g2 = torch.empty_like(xyz1), torch.empty_like(xyz2)
    _lib.check(lib.sv_chamfer_backward(_lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(idx1.contiguous()), _lib.ptr(idx2.contiguous()),
                                       _lib.ptr(grad_dist1.contiguous().float()), _lib.ptr(grad_dist2.contiguous().float()), B, n, m,
                                       _lib.ptr(g1), _lib.ptr(g2), _lib.stream()), "sv_chamfer_backward")
    return [g1, g2]
