"""roiaware_pool3d ops with the reference's names and signatures (detector3d/pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py:9-107),
on libseevcn_hip.so.  `points_in_boxes_cpu` keeps its name and numpy/torch-CPU interface but also runs on the GPU (there is no CPU
path in this build)."""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from .... import _lib


def points_in_boxes_cpu(points, boxes, device='cuda'):
    """points (num_points,3), boxes (N,7) -> point_indices (N,num_points) int32 0/1 (box test with margin 1e-2, :9-25)."""
    assert boxes.shape[1] == 7
    assert points.shape[1] == 3
    is_numpy = isinstance(points, np.ndarray)
    lib = _lib.load()
    p = (torch.from_numpy(points) if is_numpy else points).float().contiguous().to(device)
    b = (torch.from_numpy(boxes) if isinstance(boxes, np.ndarray) else boxes).float().contiguous().to(device)
    out = torch.zeros((b.shape[0], p.shape[0]), dtype=torch.int32, device=p.device)
    _lib.check(lib.sv_points_in_boxes_matrix(_lib.ptr(b) if b.numel() else None, _lib.ptr(p) if p.numel() else None, b.shape[0], p.shape[0],
                                             _lib.ptr(out) if out.numel() else None, _lib.stream()), "sv_points_in_boxes_matrix")
    out = out.cpu()
    return out.numpy() if is_numpy else out


def points_in_boxes_gpu(points, boxes):
    """points (B,M,3), boxes (B,T,7) -> (B,M) int32 box index of each point, background = -1"""
    assert boxes.shape[0] == points.shape[0]
    assert boxes.shape[2] == 7 and points.shape[2] == 3
    lib = _lib.load()
    _lib.require_cuda(points, boxes)
    batch_size, num_points, _ = points.shape
    p = points.contiguous().float()
    b = boxes.contiguous().float()
    out = torch.full((batch_size, num_points), -1, dtype=torch.int32, device=points.device)
    rc = lib.sv_points_in_boxes(_lib.ptr(b) if b.numel() else None, _lib.ptr(p) if p.numel() else None, batch_size, b.shape[1], num_points,
                                _lib.ptr(out) if out.numel() else None, _lib.stream())
    _lib.check(rc, "sv_points_in_boxes")
    return out


class RoIAwarePool3d(nn.Module):
    def __init__(self, out_size, max_pts_each_voxel=128):
        super().__init__()
        self.out_size = out_size
        self.max_pts_each_voxel = max_pts_each_voxel

    def forward(self, rois, pts, pts_feature, pool_method='max'):
        assert pool_method in ['max', 'avg']
        return RoIAwarePool3dFunction.apply(rois, pts, pts_feature, self.out_size, self.max_pts_each_voxel, pool_method)


class RoIAwarePool3dFunction(Function):
    @staticmethod
    def forward(ctx, rois, pts, pts_feature, out_size, max_pts_each_voxel, pool_method):
        """rois (N,7), pts (npoints,3), pts_feature (npoints,C) -> pooled_features (N,out_x,out_y,out_z,C)"""
        assert rois.shape[1] == 7 and pts.shape[1] == 3
        if isinstance(out_size, int):
            out_x = out_y = out_z = out_size
        else:
            assert len(out_size) == 3
            for k in range(3):
                assert isinstance(out_size[k], int)
            out_x, out_y, out_z = out_size
        lib = _lib.load()
        _lib.require_cuda(rois, pts, pts_feature)
        rois, pts, pts_feature = rois.float().contiguous(), pts.float().contiguous(), pts_feature.float().contiguous()
        num_rois, num_channels, num_pts = rois.shape[0], pts_feature.shape[-1], pts.shape[0]
        pooled_features = pts_feature.new_zeros((num_rois, out_x, out_y, out_z, num_channels))
        argmax = pts_feature.new_zeros((num_rois, out_x, out_y, out_z, num_channels), dtype=torch.int)
        pts_idx_of_voxels = pts_feature.new_zeros((num_rois, out_x, out_y, out_z, max_pts_each_voxel), dtype=torch.int)
        pool_method = {'max': 0, 'avg': 1}[pool_method]
        scratch = _lib.workspace.scratch("roiaware_pool3d", max(lib.sv_roiaware_pool3d_scratch_bytes(num_rois, num_pts), 4), rois.device)
        _lib.check(lib.sv_roiaware_pool3d_forward(_lib.ptr(rois), _lib.ptr(pts), _lib.ptr(pts_feature), num_rois, num_pts, num_channels, out_x,
                                                  out_y, out_z, int(max_pts_each_voxel), pool_method, _lib.ptr(scratch), _lib.ptr(argmax),
                                                  _lib.ptr(pts_idx_of_voxels), _lib.ptr(pooled_features), _lib.stream()),
                   "sv_roiaware_pool3d_forward")
        ctx.roiaware_pool3d_for_backward = (pts_idx_of_voxels, argmax, pool_method, num_pts, num_channels)
        return pooled_features

    @staticmethod
    def backward(ctx, grad_out):
        pts_idx_of_voxels, argmax, pool_method, num_pts, num_channels = ctx.roiaware_pool3d_for_backward
        lib = _lib.load()
        grad_out = grad_out.float().contiguous()
        grad_in = grad_out.new_zeros((num_pts, num_channels))
        n, ox, oy, oz, mp = pts_idx_of_voxels.shape
        _lib.check(lib.sv_roiaware_pool3d_backward(_lib.ptr(pts_idx_of_voxels), _lib.ptr(argmax), _lib.ptr(grad_out), n, ox, oy, oz, num_channels,
                                                   mp, pool_method, _lib.ptr(grad_in), _lib.stream()), "sv_roiaware_pool3d_backward")
        return None, None, grad_in, None, None, None
