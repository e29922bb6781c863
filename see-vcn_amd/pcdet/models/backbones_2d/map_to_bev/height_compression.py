"""HeightCompression (reference backbones_2d/map_to_bev/height_compression.py:4-26): the sparse backbone output becomes the BEV
image by stacking its depth slices into channels.  `.dense()` is the single-pass HIP scatter sv_sparse_to_dense."""
import torch.nn as nn


def _cfg(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = _cfg(model_cfg, 'NUM_BEV_FEATURES')

    def forward(self, batch_dict):
        volume = batch_dict['encoded_spconv_tensor'].dense()            # (N, C, D, H, W)
        batch_dict['spatial_features'] = volume.flatten(1, 2)           # (N, C*D, H, W): a view, channel c*D + d
        batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
        return batch_dict
