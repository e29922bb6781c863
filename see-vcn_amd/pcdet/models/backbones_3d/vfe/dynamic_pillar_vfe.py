import torch
import torch.nn as nn

from .....spconv import norm as fused_norm
from ....ops import voxel_ops
from ....utils.common_utils import cfg_get
from .vfe_template import VFETemplate


class PFNLayerV2(nn.Module):
    """Linear(+BN eps 1e-3)+ReLU over points, max over each pillar's points (reference dynamic_pillar_vfe.py:14-46);
    torch_scatter.scatter_max -> torch.scatter_reduce('amax') (every pillar has at least one point)."""

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
        self.relu = nn.ReLU()

    def forward(self, inputs, unq_inv, num_pillars):
        x = self.linear(inputs)
        if self.use_norm and fused_norm.fusable(self.norm, x):
            x = fused_norm.batch_norm_relu(self.norm, x, True)
        else:
            x = self.relu(self.norm(x) if self.use_norm else x)
        index = unq_inv.view(-1, 1).expand(-1, x.shape[1])
        x_max = x.new_zeros((num_pillars, x.shape[1])).scatter_reduce(0, index, x, reduce='amax', include_self=False)
        if self.last_vfe:
            return x_max
        return torch.cat([x, x_max[unq_inv, :]], dim=1)


class DynamicPillarVFE(VFETemplate):
    """Drop-in for the reference DynamicPillarVFE (backbones_3d/vfe/dynamic_pillar_vfe.py:49-142): same constructor keywords,
    batch_dict keys ('pillar_features' (V,C), 'voxel_coords' (V,4) [b,0,y,x]) and pillar order (ascending b*X*Y + x*Y + y).
    torch.unique + scatter_mean are one sv_voxelize_dynamic call over an (X, Y, 1) grid whose single z cell is unbounded (the
    reference does not filter on z): that kernel's key ((b*X + x)*Y + y)*Z + z is then exactly the reference's merge_coords."""

    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
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
        self.pfn_layers = nn.ModuleList([PFNLayerV2(num_filters[i], num_filters[i + 1], self.use_norm, last_layer=(i >= len(num_filters) - 2))
                                         for i in range(len(num_filters) - 1)])
        self.voxel_x, self.voxel_y, self.voxel_z = (float(v) for v in voxel_size)
        self.x_offset = self.voxel_x / 2 + float(point_cloud_range[0])
        self.y_offset = self.voxel_y / 2 + float(point_cloud_range[1])
        self.z_offset = self.voxel_z / 2 + float(point_cloud_range[2])
        self.grid_size = [int(g) for g in grid_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def forward(self, batch_dict, **kwargs):
        points = batch_dict['points']  # (batch_idx, x, y, z, i, e)
        r = self.point_cloud_range
        with torch.no_grad():
            points_mean, coords, p2v = voxel_ops.voxelize_dynamic(
                points, [r[0], r[1], -1e30, r[3], r[4], 1e30], [self.voxel_x, self.voxel_y, 2e30], [self.grid_size[0], self.grid_size[1], 1],
                batch_dict['batch_size'], num_features=3, return_point_to_voxel=True)
        mask = p2v >= 0
        points = points[mask]
        unq_inv = p2v[mask].long()
        points_xyz = points[:, [1, 2, 3]].contiguous()
        f_cluster = points_xyz - points_mean[unq_inv, :]
        vc = coords[unq_inv]                                           # per point [b, 0, y, x]
        f_center = torch.zeros_like(points_xyz)
        f_center[:, 0] = points_xyz[:, 0] - (vc[:, 3].to(points_xyz.dtype) * self.voxel_x + self.x_offset)
        f_center[:, 1] = points_xyz[:, 1] - (vc[:, 2].to(points_xyz.dtype) * self.voxel_y + self.y_offset)
        f_center[:, 2] = points_xyz[:, 2] - self.z_offset
        features = [points[:, 1:], f_cluster, f_center] if self.use_absolute_xyz else [points[:, 4:], f_cluster, f_center]
        if self.with_distance:
            features.append(torch.norm(points[:, 1:4], 2, dim=1, keepdim=True))
        features = torch.cat(features, dim=-1)
        for pfn in self.pfn_layers:
            features = pfn(features, unq_inv, coords.shape[0])
        batch_dict['pillar_features'] = features
        batch_dict['voxel_coords'] = coords.contiguous()                # [b, 0, y, x]
        return batch_dict
