"""Chamfer distance modules with the reference's names (see/surface_completion/models/vcn/extensions/chamfer_dist/__init__.py:13-102)
over sv_chamfer_forward / sv_chamfer_backward."""
import torch

from .... import _lib


class ChamferFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        lib = _lib.load()
        _lib.require_cuda(xyz1, xyz2)
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        B, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
        dev = xyz1.device
        dist1, dist2 = torch.empty((B, n), dtype=torch.float32, device=dev), torch.empty((B, m), dtype=torch.float32, device=dev)
        idx1, idx2 = torch.empty((B, n), dtype=torch.int32, device=dev), torch.empty((B, m), dtype=torch.int32, device=dev)
        _lib.check(lib.sv_chamfer_forward(_lib.ptr(xyz1), _lib.ptr(xyz2), B, n, m, _lib.ptr(dist1), _lib.ptr(dist2), _lib.ptr(idx1), _lib.ptr(idx2),
                                          _lib.stream()), "sv_chamfer_forward")
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, grad_dist1, grad_dist2):
        lib = _lib.load()
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        B, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
        g1, g2 = torch.empty_like(xyz1), torch.empty_like(xyz2)
        _lib.check(lib.sv_chamfer_backward(_lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(idx1), _lib.ptr(idx2), _lib.ptr(grad_dist1.contiguous().float()),
                                           _lib.ptr(grad_dist2.contiguous().float()), B, n, m, _lib.ptr(g1), _lib.ptr(g2), _lib.stream()),
                   "sv_chamfer_backward")
        return g1, g2


def _strip_zeros(xyz1, xyz2, ignore_zeros):
    if xyz1.size(0) == 1 and ignore_zeros:
        xyz1 = xyz1[torch.sum(xyz1, dim=2).ne(0)].unsqueeze(dim=0)
        xyz2 = xyz2[torch.sum(xyz2, dim=2).ne(0)].unsqueeze(dim=0)
    return xyz1, xyz2


class ChamferDistanceL2(torch.nn.Module):
    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def forward(self, xyz1, xyz2):
        xyz1, xyz2 = _strip_zeros(xyz1, xyz2, self.ignore_zeros)
        dist1, dist2 = ChamferFunction.apply(xyz1, xyz2)
        return torch.mean(dist1) + torch.mean(dist2)


class ChamferDistanceL2_split(torch.nn.Module):
    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def forward(self, xyz1, xyz2):
        xyz1, xyz2 = _strip_zeros(xyz1, xyz2, self.ignore_zeros)
        dist1, dist2 = ChamferFunction.apply(xyz1, xyz2)
        return torch.mean(dist1), torch.mean(dist2)


class ChamferDistanceL1(torch.nn.Module):
    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def forward(self, xyz1, xyz2):
        xyz1, xyz2 = _strip_zeros(xyz1, xyz2, self.ignore_zeros)
        dist1, dist2 = ChamferFunction.apply(xyz1, xyz2)
        return (torch.mean(torch.sqrt(dist1)) + torch.mean(torch.sqrt(dist2))) / 2
