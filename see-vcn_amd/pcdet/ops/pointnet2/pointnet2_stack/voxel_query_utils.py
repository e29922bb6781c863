"""voxel_query / VoxelQueryAndGrouping with the reference's names (detector3d/pcdet/ops/pointnet2/pointnet2_stack/voxel_query_utils.py)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_stack_cuda as pointnet2
from . import pointnet2_utils


class VoxelQuery(Function):
    @staticmethod
    def forward(ctx, max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
        """new_coords (M,4) [b,z,y,x] voxel of each query, point_indices (B,Z,Y,X) voxel -> point row (or -1).
        Returns idx (M,nsample) int32 and empty_ball_mask (M) bool (voxel_query_utils.py:13-45)."""
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        assert new_coords.is_contiguous()
        assert point_indices.is_contiguous()
        M = new_coords.shape[0]
        B, Z, Y, X = point_indices.shape
        idx = torch.zeros((M, nsample), dtype=torch.int32, device=xyz.device)
        z_range, y_range, x_range = max_range
        pointnet2.voxel_query_wrapper(M, Z, Y, X, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, point_indices, idx)
        empty_ball_mask = (idx[:, 0] == -1)
        idx[empty_ball_mask] = 0
        return idx, empty_ball_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None, None


voxel_query = VoxelQuery.apply


class VoxelQueryAndGrouping(nn.Module):
    def __init__(self, max_range, radius, nsample):
        super().__init__()
        self.max_range, self.radius, self.nsample = max_range, radius, nsample

    def forward(self, new_coords, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, voxel2point_indices):
        """voxel_query_utils.py:62-100 -> grouped_features (M,C,nsample), grouped_xyz (M,3,nsample), empty_ball_mask (M)"""
        assert xyz.shape[0] == xyz_batch_cnt.sum(), 'xyz: %s, xyz_batch_cnt: %s' % (str(xyz.shape), str(new_xyz_batch_cnt))
        assert new_coords.shape[0] == new_xyz_batch_cnt.sum(), \
            'new_coords: %s, new_xyz_batch_cnt: %s' % (str(new_coords.shape), str(new_xyz_batch_cnt))
        batch_size = xyz_batch_cnt.shape[0]
        idx1, empty_ball_mask1 = voxel_query(self.max_range, self.radius, self.nsample, xyz, new_xyz, new_coords, voxel2point_indices)
        idx1 = idx1.view(batch_size, -1, self.nsample)
        count = 0
        for bs_idx in range(batch_size):
            idx1[bs_idx] -= count
            count += xyz_batch_cnt[bs_idx]
        idx1 = idx1.view(-1, self.nsample)
        idx1[empty_ball_mask1] = 0
        idx = idx1
        empty_ball_mask = empty_ball_mask1
        grouped_xyz = pointnet2_utils.grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        grouped_features = pointnet2_utils.grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        return grouped_features, grouped_xyz, empty_ball_mask
