import torch.nn as nn


class HeightCompression(nn.Module):
    """Drop-in for the reference HeightCompression (backbones_2d/map_to_bev/height_compression.py:4-26):
    .dense() runs the single-pass HIP scatter sv_sparse_to_dense, then (N,C,D,H,W) -> (N,C*D,H,W)."""

    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES if hasattr(self.model_cfg, 'NUM_BEV_FEATURES') \
            else self.model_cfg['NUM_BEV_FEATURES']

    def forward(self, batch_dict):
        encoded_spconv_tensor = batch_dict['encoded_spconv_tensor']
        spatial_features = encoded_spconv_tensor.dense()
        N, C, D, H, W = spatial_features.shape
        spatial_features = spatial_features.view(N, C * D, H, W)
        batch_dict['spatial_features'] = spatial_features
        batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
        return batch_dict
