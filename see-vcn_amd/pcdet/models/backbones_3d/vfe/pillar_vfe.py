import torch
import torch.nn as nn
import torch.nn.functional as F

from ....ops import voxel_ops
from ....utils.common_utils import cfg_get
from .vfe_template import VFETemplate


class PFNLayer(nn.Module):
    """Linear(+BatchNorm1d eps 1e-3)+ReLU+max over the pillar's points; same parameters as the reference PFNLayer
    (backbones_3d/vfe/pillar_vfe.py:8-49)."""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.use_norm = use_norm
        if not self.last_vfe:
            out_channels = out_channels // 2
        if self.use_norm:
            self.linear = nn.Linear(in_channels, out_channels, bias=False)
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        else:
            self.linear = nn.Linear(in_channels, out_channels, bias=True)

    def forward(self, inputs):
        x = self.linear(inputs)
        x = self.norm(x.permute(0, 2, 1)).permute(0, 2, 1) if self.use_norm else x
        x = F.relu(x)
        x_max = torch.max(x, dim=1, keepdim=True)[0]
        if self.last_vfe:
            return x_max
        return torch.cat([x, x_max.repeat(1, inputs.shape[1], 1)], dim=2)


class PillarVFE(VFETemplate):
    """Drop-in for the reference PillarVFE (pillar_vfe.py:52-123): the per-point decoration (cluster / centre offsets,
    padding mask) runs in the HIP kernel sv_pillar_decorate; the 10->64 Linear + BN + ReLU + max stay library ops."""

    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.use_norm = cfg_get(model_cfg, 'USE_NORM')
        self.with_distance = cfg_get(model_cfg, 'WITH_DISTANCE')
        self.use_absolute_xyz = cfg_get(model_cfg, 'USE_ABSLOTE_XYZ')
        num_point_features += 6 if self.use_absolute_xyz else 3
        if self.with_distance:
            num_point_features += 1
        self.num_filters = cfg_get(model_cfg, 'NUM_FILTERS')
        assert len(self.num_filters) > 0
        num_filters = [num_point_features] + list(self.num_filters)
        self.pfn_layers = nn.ModuleList([
            PFNLayer(num_filters[i], num_filters[i + 1], self.use_norm, last_layer=(i >= len(num_filters) - 2))
            for i in range(len(num_filters) - 1)])
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def forward(self, batch_dict, **kwargs):
        features = voxel_ops.pillar_decorate(batch_dict['voxels'], batch_dict['voxel_num_points'], batch_dict['voxel_coords'],
                                             self.voxel_size, self.point_cloud_range, self.use_absolute_xyz, self.with_distance)
        for pfn in self.pfn_layers:
            features = pfn(features)
        batch_dict['pillar_features'] = features.squeeze()
        return batch_dict
