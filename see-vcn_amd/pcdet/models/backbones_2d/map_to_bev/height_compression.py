"""HeightCompression (reference backbones_2d/map_to_bev/height_compression.py:4-26): the sparse backbone output becomes the BEV
image by stacking its depth slices into channels.  `.dense()` is the single-pass HIP scatter sv_sparse_to_dense."""
import os

import torch.nn as nn

# memory format of spatial_features: follows SEEVCN_BEV_FORMAT (auto: channels_last from 8 scenes per batch on, like BaseBEVBackbone)
CHANNELS_LAST = os.environ.get("SEEVCN_BEV_FORMAT", "auto")


def _cfg(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = _cfg(model_cfg, 'NUM_BEV_FEATURES')
        # set by the detector when a BaseBEVBackbone reads spatial_features (Detector3DTemplate.build_backbone_2d): only then does a channels_last
        # volume save anything -- any other consumer gets the reference's (N, C*D, H, W) contiguous view
        self.feeds_bev_backbone = False

    def forward(self, batch_dict):
        enc = batch_dict['encoded_spconv_tensor']
        feats = None
        if self.feeds_bev_backbone and (CHANNELS_LAST == "nhwc" or (CHANNELS_LAST == "auto" and enc.batch_size >= 8)):
            # the 2-D backbone behind runs channels_last at this batch size (base_bev_backbone.py, same rule): the volume is written in that order, same
            # values at the same (n, c * D + d, y, x) -- a (N, C*D, H, W) tensor with channels_last strides instead of a 577 MB copy each way
            from .....spconv import functional as Fsp
            feats = Fsp.sparse_to_dense_channels_last(enc.features, enc.indices, enc.batch_size, enc.spatial_shape)
        if feats is None:
            volume = enc.dense()                                        # (N, C, D, H, W)
            feats = volume.flatten(1, 2)                                # (N, C*D, H, W): a view, channel c*D + d
        batch_dict['spatial_features'] = feats
        batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
        return batch_dict
