"""Chamfer distance modules with the reference's names (see/surface_completion/models/vcn/extensions/chamfer_dist/__init__.py:13-102)
over chamfer.forward / chamfer.backward (chamfer.py -> sv_chamfer_forward / sv_chamfer_backward)."""
import torch

from . import chamfer


class ChamferFunction(torch.autograd.Function):
    """Same body as the reference's (chamfer_dist/__init__.py:13-25) over the pybind-level module `chamfer`."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        dist1, dist2, idx1, idx2 = chamfer.forward(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, grad_dist1, grad_dist2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        grad_xyz1, grad_xyz2 = chamfer.backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2)
        return grad_xyz1, grad_xyz2


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
