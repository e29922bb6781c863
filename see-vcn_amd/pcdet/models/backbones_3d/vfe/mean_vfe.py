from ....ops import voxel_ops
from .vfe_template import VFETemplate


class MeanVFE(VFETemplate):
    """Drop-in for the reference MeanVFE (backbones_3d/vfe/mean_vfe.py:6-31) on the HIP kernel sv_mean_vfe."""

    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        batch_dict['voxel_features'] = voxel_ops.mean_vfe(batch_dict['voxels'], batch_dict['voxel_num_points'])
        return batch_dict
