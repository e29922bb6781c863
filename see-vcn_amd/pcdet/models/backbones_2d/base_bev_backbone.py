import os

import numpy as np
import torch
import torch.nn as nn

from ...utils.common_utils import cfg_get


def _conv_bn_relu(c_in, c_out, k, stride, padding):
    return [nn.Conv2d(c_in, c_out, kernel_size=k, stride=stride, padding=padding, bias=False),
            nn.BatchNorm2d(c_out, eps=1e-3, momentum=0.01), nn.ReLU()]


# Memory format of the dense 2-D part (MIOpen's role is cuDNN's): 'auto' = channels_last (NHWC) from 8 scenes per batch on, NCHW below.  Measured on one
# MI355X, forward + backward of this module (tools/bev_format_ab.py, profiles/r03_bev_format_ab.txt): 16 x 256 x 200 x 176 (SECOND) 74.0 ms NCHW / 68.9 ms
# NHWC; 4 x 256 x 188 x 188 (PV-RCNN) 18.3 / 23.3 ms -- MIOpen picks NHWC implicit-GEMM kernels for the weight gradients either way and transposes around
# them when fed NCHW; at small batch its NCHW Winograd forward wins by more than the transposes cost.  SEEVCN_BEV_FORMAT=nchw|nhwc forces one.
BEV_FORMAT = os.environ.get("SEEVCN_BEV_FORMAT", "auto")


class BaseBEVBackbone(nn.Module):
    """Drop-in for the reference BaseBEVBackbone (backbones_2d/base_bev_backbone.py:6-112): same config keys, the same
    `blocks` / `deblocks` module lists (state_dict keys) and batch_dict keys.  Dense 2-D convolutions stay on MIOpen
    (on the path, not a hand-kernel target)."""

    def __init__(self, model_cfg, input_channels):
        super().__init__()
        self.model_cfg = model_cfg
        layer_nums = cfg_get(model_cfg, 'LAYER_NUMS', None) or []
        layer_strides = cfg_get(model_cfg, 'LAYER_STRIDES', None) or []
        num_filters = cfg_get(model_cfg, 'NUM_FILTERS', None) or []
        assert len(layer_nums) == len(layer_strides) == len(num_filters)
        upsample_strides = cfg_get(model_cfg, 'UPSAMPLE_STRIDES', None) or []
        num_upsample_filters = cfg_get(model_cfg, 'NUM_UPSAMPLE_FILTERS', None) or []
        assert len(upsample_strides) == len(num_upsample_filters)
        num_levels = len(layer_nums)
        c_in_list = [input_channels, *num_filters[:-1]]
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for idx in range(num_levels):
            layers = [nn.ZeroPad2d(1)] + _conv_bn_relu(c_in_list[idx], num_filters[idx], 3, layer_strides[idx], 0)
            for _ in range(layer_nums[idx]):
                layers += _conv_bn_relu(num_filters[idx], num_filters[idx], 3, 1, 1)
            self.blocks.append(nn.Sequential(*layers))
            if len(upsample_strides) > 0:
                stride = upsample_strides[idx]
                if stride >= 1:
                    up = nn.ConvTranspose2d(num_filters[idx], num_upsample_filters[idx], stride, stride=stride, bias=False)
                else:
                    s = int(np.round(1 / stride))
                    up = nn.Conv2d(num_filters[idx], num_upsample_filters[idx], s, stride=s, bias=False)
                self.deblocks.append(nn.Sequential(up, nn.BatchNorm2d(num_upsample_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()))
        c_in = sum(num_upsample_filters)
        if len(upsample_strides) > num_levels:
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(c_in, c_in, upsample_strides[-1], stride=upsample_strides[-1], bias=False),
                nn.BatchNorm2d(c_in, eps=1e-3, momentum=0.01), nn.ReLU()))
        self.num_bev_features = c_in

    def forward(self, data_dict):
        spatial_features = data_dict['spatial_features']
        ups = []
        x = spatial_features
        if x.is_cuda and (BEV_FORMAT == "nhwc" or (BEV_FORMAT == "auto" and x.shape[0] >= 8)):
            if not getattr(self, "_nhwc", False):
                self.to(memory_format=torch.channels_last)           # parameters re-laid once; gradients follow their parameters' layout
                self._nhwc = True
            x = x.contiguous(memory_format=torch.channels_last)
        for i in range(len(self.blocks)):
            x = self.blocks[i](x)
            stride = int(spatial_features.shape[2] / x.shape[2])
            data_dict['spatial_features_%dx' % stride] = x
            ups.append(self.deblocks[i](x) if len(self.deblocks) > 0 else x)
        if len(ups) > 1:
            x = torch.cat(ups, dim=1)
        elif len(ups) == 1:
            x = ups[0]
        if len(self.deblocks) > len(self.blocks):
            x = self.deblocks[-1](x)
        data_dict['spatial_features_2d'] = x
        return data_dict
