import torch

from ....ops import voxel_ops
from .vfe_template import VFETemplate


class DynamicMeanVFE(VFETemplate):
    """Drop-in for the reference DynamicMeanVFE (backbones_3d/vfe/dynamic_mean_vfe.py:14-76).

    Same constructor keywords and batch_dict keys; the torch.unique + torch_scatter pipeline is replaced by
    sv_voxelize_dynamic (bitmap rank index in HBM, no sort).  voxel_coords come out bit-identical and in
    the same (ascending key) order; voxel_features are sum/count in fp32.
    """

    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features
        self.grid_size = [int(g) for g in grid_size]
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]

    def get_output_feature_dim(self):
        return self.num_point_features

    @torch.no_grad()
    def forward(self, batch_dict, lazy_count=False, **kwargs):
        """lazy_count (seevcn extension, default off = the reference's contract): no device -> host read here -- voxel_features / voxel_coords keep
        their CAPACITY rows (one per point) and batch_dict['voxel_count_device'] holds the true count on the device; the consumer
        (spconv.prebuild_rulebooks(..., n0_dev=)) reads it together with its own counts and narrows the tensors."""
        points = batch_dict['points']  # (batch_idx, x, y, z, i, e)
        if lazy_count:
            feats, coords, _, nvox = voxel_ops.voxelize_dynamic(points, self.point_cloud_range, self.voxel_size, self.grid_size, batch_dict['batch_size'], sync=False)
            batch_dict['voxel_features'], batch_dict['voxel_coords'], batch_dict['voxel_count_device'] = feats, coords, nvox
            return batch_dict
        feats, coords, _ = voxel_ops.voxelize_dynamic(
            points, self.point_cloud_range, self.voxel_size, self.grid_size, batch_dict['batch_size'])
        batch_dict['voxel_features'] = feats.contiguous()
        batch_dict['voxel_coords'] = coords.contiguous()
        return batch_dict
