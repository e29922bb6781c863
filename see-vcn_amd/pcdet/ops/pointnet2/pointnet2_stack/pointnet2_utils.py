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


class SAScaleTrain(Function):
    """One radius scale of StackSAModuleMSG in TRAINING mode as one autograd node over the hand-written MFMA kernels
    (sv_sa_train_forward / sv_sa_train_backward, csrc/set_abstraction_train.hip): gather -> Conv2d 1x1 -> BatchNorm2d(batch statistics) ->
    ReLU, twice -> max over nsample (pointnet2_modules.py:96-110), without the (M, C+3, nsample) tensor; running statistics and
    num_batches_tracked of the two norms are updated like nn.BatchNorm2d does.  idx / row_start as ball_query_wrapper leaves them."""

    @staticmethod
    def forward(ctx, xyz, features, new_xyz, idx, row_start, w1, g1, b1, w2, g2, b2, rm1, rv1, nbt1, rm2, rv2, nbt2, momentum, eps):
        from ..... import _lib
        lib = _lib.load()
        dev = xyz.device
        M, ns = idx.shape
        C = 0 if features is None else features.shape[1]
        C1, C2 = w1.shape[0], w2.shape[0]
        w1c, w2c = w1.detach().reshape(C1, C + 3).contiguous(), w2.detach().reshape(C2, C1).contiguous()
        R = M * ns
        f32 = dict(dtype=torch.float32, device=dev)
        N = xyz.shape[0]
        proj = torch.empty((N, C1), **f32) if C else None                     # layer 1's feature part, once per support point (work buffer)
        z1, z2 = torch.empty((R, C1), **f32), torch.empty((R, C2), **f32)
        stats = torch.empty((2 * C1 + 2 * C2,), **f32)
        sm1, si1, sm2, si2 = stats[:C1], stats[C1:2 * C1], stats[2 * C1:2 * C1 + C2], stats[2 * C1 + C2:]
        sel, aux, out = torch.empty((M, C2), **f32), torch.empty((M, C2), **f32), torch.empty((M, C2), **f32)
        arg, aux_arg = torch.empty((M, C2), dtype=torch.uint8, device=dev), torch.empty((M, C2), dtype=torch.uint8, device=dev)
        scratch = _lib.workspace.scratch("sa_train", lib.sv_sa_train_scratch_bytes(C, C1, C2), dev)
        _lib.check(lib.sv_sa_train_forward(_lib.ptr(xyz), _lib.ptr(features) if C else None, _lib.ptr(new_xyz), _lib.ptr(idx), _lib.ptr(row_start), M, N, C, ns,
                                           _lib.ptr(w1c), _lib.ptr(g1.detach()), _lib.ptr(b1.detach()), _lib.ptr(rm1), _lib.ptr(rv1), _lib.ptr(nbt1), C1,
                                           _lib.ptr(w2c), _lib.ptr(g2.detach()), _lib.ptr(b2.detach()), _lib.ptr(rm2), _lib.ptr(rv2), _lib.ptr(nbt2), C2,
                                           float(momentum), float(eps), _lib.ptr(scratch), _lib.ptr(proj), _lib.ptr(z1), _lib.ptr(z2), _lib.ptr(sm1), _lib.ptr(si1),
                                           _lib.ptr(sm2), _lib.ptr(si2), _lib.ptr(sel), _lib.ptr(aux), _lib.ptr(arg), _lib.ptr(aux_arg), _lib.ptr(out),
                                           _lib.stream()), "sv_sa_train_forward")
        ctx.save_for_backward(xyz, features, new_xyz, idx, row_start, w1c, g1, b1, w2c, g2, b2, z1, z2, stats, sel, arg, out)
        ctx.w_shapes = (w1.shape, w2.shape)
        ctx.mark_non_differentiable(idx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from ..... import _lib
        lib = _lib.load()
        xyz, features, new_xyz, idx, row_start, w1c, g1, b1, w2c, g2, b2, z1, z2, stats, sel, arg, out = ctx.saved_tensors
        dev = xyz.device
        M, ns = idx.shape
        C = 0 if features is None else features.shape[1]
        N = xyz.shape[0]
        C1, C2 = w1c.shape[0], w2c.shape[0]
        sm1, si1, sm2, si2 = stats[:C1], stats[C1:2 * C1], stats[2 * C1:2 * C1 + C2], stats[2 * C1 + C2:]
        f32 = dict(dtype=torch.float32, device=dev)
        dy1, aux = torch.empty((M * ns, C1), **f32), torch.empty((M, C2), **f32)
        scatter = torch.empty((N, C1), **f32) if C else None
        gf = torch.empty((N, C), **f32) if C and ctx.needs_input_grad[1] else None
        gw1, gw2 = torch.empty((C1, C + 3), **f32), torch.empty((C2, C1), **f32)
        gbn = torch.empty((2 * C1 + 2 * C2,), **f32)
        dg1, db1, dg2, db2 = gbn[:C1], gbn[C1:2 * C1], gbn[2 * C1:2 * C1 + C2], gbn[2 * C1 + C2:]
        scratch = _lib.workspace.scratch("sa_train", lib.sv_sa_train_scratch_bytes(C, C1, C2), dev)
        _lib.check(lib.sv_sa_train_backward(_lib.ptr(xyz), _lib.ptr(features) if C else None, _lib.ptr(new_xyz), _lib.ptr(idx), _lib.ptr(row_start), M, N, C,
                                            ns, _lib.ptr(w1c), _lib.ptr(g1.detach()), _lib.ptr(b1.detach()), C1, _lib.ptr(w2c), _lib.ptr(g2.detach()),
                                            _lib.ptr(b2.detach()), C2, _lib.ptr(z1), _lib.ptr(z2), _lib.ptr(sm1), _lib.ptr(si1), _lib.ptr(sm2), _lib.ptr(si2),
                                            _lib.ptr(sel), _lib.ptr(arg), _lib.ptr(out), _lib.ptr(grad_out.contiguous()), _lib.ptr(scratch), _lib.ptr(dy1),
                                            _lib.ptr(aux), _lib.ptr(scatter), _lib.ptr(gf), _lib.ptr(gw1), _lib.ptr(gw2), _lib.ptr(dg1), _lib.ptr(db1), _lib.ptr(dg2),
                                            _lib.ptr(db2), _lib.stream()), "sv_sa_train_backward")
        s1, s2 = ctx.w_shapes
        return (None, gf, None, None, None, gw1.view(s1), dg1, db1, gw2.view(s2), dg2, db2, None, None, None, None, None, None, None, None)


sa_scale_train = SAScaleTrain.apply


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
