"""Autograd functions / modules of the reference's pointnet2_batch/pointnet2_utils.py (same names), on the HIP wrappers."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_batch_cuda as pointnet2


class _IndexOp:
    """Index-producing ops carry no gradient: plain functions run without autograd, exposed with the `.apply` surface the
    reference's autograd.Function classes have (pointnet2_utils.py:10-40,87-123,228-260)."""

    @classmethod
    def apply(cls, *args):
        with torch.no_grad():
            return cls.run(*args)


class FarthestPointSampling(_IndexOp):
    @staticmethod
    def run(xyz, npoint):
        """xyz (B,N,3) -> (B,npoint) int32 indices; start index 0, the reference kernel's tie rule"""
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        idx = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        scratch = torch.empty((B, N), dtype=torch.float32, device=xyz.device)
        pointnet2.farthest_point_sampling_wrapper(B, N, npoint, xyz, scratch, idx)
        return idx


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous() and idx.is_contiguous()
        B, npoint = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, npoint), dtype=torch.float32, device=features.device)
        pointnet2.gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.empty((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2.gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(_IndexOp):
    @staticmethod
    def run(unknown, known):
        """unknown (B,n,3), known (B,m,3) -> (distances (B,n,3), indices (B,n,3)) of the three nearest known points"""
        assert unknown.is_contiguous() and known.is_contiguous()
        B, n, _ = unknown.size()
        d2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknown.device)
        idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknown.device)
        pointnet2.three_nn_wrapper(B, n, known.size(1), unknown, known, d2, idx)
        return d2.sqrt_(), idx


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = torch.empty((B, c, n), dtype=torch.float32, device=features.device)
        pointnet2.three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.empty((B, c, m), dtype=torch.float32, device=grad_out.device)
        pointnet2.three_interpolate_grad_wrapper(B, c, n, m, grad_out.contiguous(), idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous() and idx.is_contiguous()
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, nfeatures, nsample), dtype=torch.float32, device=features.device)
        pointnet2.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.empty((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(_IndexOp):
    @staticmethod
    def run(radius, nsample, xyz, new_xyz):
        """first nsample points of xyz (B,N,3) within radius of each new_xyz (B,npoint,3) -> (B,npoint,nsample) int32"""
        assert new_xyz.is_contiguous() and xyz.is_contiguous()
        B, N, _ = xyz.size()
        idx = torch.zeros((B, new_xyz.size(1), nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(B, N, new_xyz.size(1), radius, nsample, new_xyz, xyz, idx)
        return idx


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        """xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) -> (B, 3+C, npoint, nsample)"""
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        return grouped_xyz


class GroupAll(nn.Module):
    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        return grouped_xyz
