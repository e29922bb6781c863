"""Autograd functions / modules with the reference's names and call signatures
(detector3d/pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py:8-188)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_stack_cuda as pointnet2


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
        assert new_xyz.is_contiguous() and new_xyz_batch_cnt.is_contiguous()
        assert xyz.is_contiguous() and xyz_batch_cnt.is_contiguous()
        B = xyz_batch_cnt.shape[0]
        M = new_xyz.shape[0]
        idx = torch.zeros((M, nsample), dtype=torch.int32, device=new_xyz.device)
        pointnet2.ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
        empty_ball_mask = (idx[:, 0] == -1)
        idx[empty_ball_mask] = 0
        ctx.mark_non_differentiable(idx)
        ctx.mark_non_differentiable(empty_ball_mask)
        return idx, empty_ball_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None


ball_query = BallQuery.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, features_batch_cnt, idx, idx_batch_cnt):
        assert features.is_contiguous() and features_batch_cnt.is_contiguous()
        assert idx.is_contiguous() and idx_batch_cnt.is_contiguous()
        assert features.shape[0] == features_batch_cnt.sum(), \
            'features: %s, features_batch_cnt: %s' % (str(features.shape), str(features_batch_cnt))
        assert idx.shape[0] == idx_batch_cnt.sum(), 'idx: %s, idx_batch_cnt: %s' % (str(idx.shape), str(idx_batch_cnt))
        M, nsample = idx.size()
        N, C = features.size()
        B = idx_batch_cnt.shape[0]
        output = torch.empty((M, C, nsample), dtype=torch.float32, device=features.device)
        pointnet2.group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, output)
        ctx.for_backwards = (B, N, idx, features_batch_cnt, idx_batch_cnt)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        B, N, idx, features_batch_cnt, idx_batch_cnt = ctx.for_backwards
        M, C, nsample = grad_out.size()
        grad_features = torch.empty((N, C), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(B, M, C, N, nsample, grad_out.contiguous(), idx, idx_batch_cnt, features_batch_cnt,
                                            grad_features)
        return grad_features, None, None, None


grouping_operation = GroupingOperation.apply


class GroupRows(Function):
    """Neighbourhoods as rows (seevcn extension): (M * nsample, 3 + C) = [xyz[j] - new_xyz[m] | features[j]], zero rows for empty balls --
    the channel-last twin of QueryAndGroup's (M, 3 + C, nsample) tensor, so that a scale's shared MLP is one GEMM over all rows.
    idx (M, nsample) int32 as ball_query_wrapper leaves it (idx[m][0] == -1 marks an empty ball); row_start (M,) first row of the query's scene."""

    @staticmethod
    def forward(ctx, xyz, features, new_xyz, idx, row_start):
        from ..... import _lib
        lib = _lib.load()
        M, ns = idx.shape
        C = 0 if features is None else features.shape[1]
        out = torch.empty((M * ns, 3 + C), dtype=torch.float32, device=xyz.device)
        _lib.check(lib.sv_group_rows_stack(M, C, ns, _lib.ptr(xyz), _lib.ptr(features) if C else None, _lib.ptr(new_xyz), _lib.ptr(idx), _lib.ptr(row_start),
                                           _lib.ptr(out), _lib.stream()), "sv_group_rows_stack")
        ctx.n = 0 if features is None else features.shape[0]
        ctx.c = C
        ctx.save_for_backward(idx, row_start)
        return out

    @staticmethod
    def backward(ctx, grad_rows):
        from ..... import _lib
        if ctx.c == 0 or not ctx.needs_input_grad[1]:
            return None, None, None, None, None
        lib = _lib.load()
        idx, row_start = ctx.saved_tensors
        M, ns = idx.shape
        grad = torch.empty((ctx.n, ctx.c), dtype=torch.float32, device=grad_rows.device)
        _lib.check(lib.sv_group_rows_grad_stack(M, ctx.c, ctx.n, ns, _lib.ptr(grad_rows.contiguous()), _lib.ptr(idx), _lib.ptr(row_start), _lib.ptr(grad),
                                                _lib.stream()), "sv_group_rows_grad_stack")
        return None, grad, None, None, None


group_rows = GroupRows.apply


class RowsLinear(Function):
    """y = x @ w.T for a tall x (R rows, R ~ 10^5..10^6, 16..259 columns).  The weight gradient dy.T @ x contracts over R into a tiny
    (C_out, C_in) result: as ONE library GEMM that is a single output tile walking all R rows (7.6 ms at R = 1.77 M on MI355X); here R is cut
    into S slices contracted by one batched GEMM and summed (split-K), S chosen so every slice still has >= 1024 rows."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ w if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            r = x.shape[0]
            s = 1
            while s < 256 and r % (2 * s) == 0 and r // (2 * s) >= 1024:
                s *= 2
            if s == 1:
                dw = dy.t() @ x
            else:
                dw = torch.bmm(dy.view(s, r // s, -1).transpose(1, 2), x.view(s, r // s, -1)).sum(dim=0)
        return dx, dw


rows_linear = RowsLinear.apply


class QueryAndGroup(nn.Module):
    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None):
        """xyz (N1+N2..,3), new_xyz (M1+M2..,3), features (N1+N2..,C) -> new_features (M1+M2.., C(+3), nsample), idx"""
        assert xyz.shape[0] == xyz_batch_cnt.sum(), 'xyz: %s, xyz_batch_cnt: %s' % (str(xyz.shape), str(new_xyz_batch_cnt))
        assert new_xyz.shape[0] == new_xyz_batch_cnt.sum(), \
            'new_xyz: %s, new_xyz_batch_cnt: %s' % (str(new_xyz.shape), str(new_xyz_batch_cnt))
        idx, empty_ball_mask = ball_query(self.radius, self.nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        grouped_xyz = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)  # (M, 3, nsample)
        grouped_xyz = grouped_xyz - new_xyz.unsqueeze(-1)
        grouped_xyz[empty_ball_mask] = 0
        if features is not None:
            grouped_features = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)  # (M, C, nsample)
            grouped_features[empty_ball_mask] = 0
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return new_features, idx


class FarthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) -> (B,npoint) int32 (reference pointnet2_utils.py:162-188)"""
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        temp = torch.empty((B, N), dtype=torch.float32, device=xyz.device) if N > 24576 else None
        pointnet2.farthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        return output

    @staticmethod
    def backward(xyz, a=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply
stack_farthest_point_sample = pointnet2.stack_farthest_point_sampling
