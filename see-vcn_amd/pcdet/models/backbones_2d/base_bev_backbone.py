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
BN_FUSED = os.environ.get("SEEVCN_BEV_BN_FUSED", "1")      # 0: channels_last activations still go through MIOpen's BatchNorm2d and torch's ReLU (A/B)


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

    @staticmethod
    def _run(seq, x, nhwc):
        """seq(x).  In channels_last an activation IS the (N H W, C) row matrix of the sparse backbone's BatchNorm kernels: every BatchNorm2d (+ ReLU behind
        it) runs as ONE fused pass pair (spconv.norm: batch statistics in fp64 over fixed-order partials, ReLU inside, the backward recomputes the branch) in
        place of MIOpen's BatchNorm + two elementwise ReLU passes.  Modules with hooks, odd channel counts or NCHW tensors go through the modules."""
        if not nhwc or BN_FUSED == "0":
            return seq(x)
        from ....spconv import norm
        mods, i = list(seq), 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            fuse = (isinstance(m, nn.BatchNorm2d) and x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and norm.channels_fusable(m.num_features)
                    and m.momentum is not None and (m.training or m.track_running_stats) and (not x.requires_grad or m.training)
                    and not (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks))
            if not fuse:
                x = m(x)
                i += 1
                continue
            relu = isinstance(nxt, nn.ReLU) and not (nxt._forward_hooks or nxt._forward_pre_hooks or nxt._backward_hooks)
            n, c, h, w = x.shape
            rows = x.permute(0, 2, 3, 1)                             # a view of a channels_last tensor; anything else is copied once
            if not rows.is_contiguous():
                rows = rows.contiguous()
            y = norm.batch_norm_relu(m, rows.reshape(-1, c), relu)
            x = y.view(n, h, w, c).permute(0, 3, 1, 2)
            i += 2 if relu else 1
        return x

    def forward(self, data_dict):
        spatial_features = data_dict['spatial_features']
        ups = []
        x = spatial_features
        nhwc = x.is_cuda and (BEV_FORMAT == "nhwc" or (BEV_FORMAT == "auto" and x.shape[0] >= 8))
        if nhwc:
            if not getattr(self, "_nhwc", False):
                self.to(memory_format=torch.channels_last)           # parameters re-laid once; gradients follow their parameters' layout
                self._nhwc = True
            x = x.contiguous(memory_format=torch.channels_last)
        for i in range(len(self.blocks)):
            x = self._run(self.blocks[i], x, nhwc)
            stride = int(spatial_features.shape[2] / x.shape[2])
            data_dict['spatial_features_%dx' % stride] = x
            ups.append(self._run(self.deblocks[i], x, nhwc) if len(self.deblocks) > 0 else x)
        if len(ups) > 1:
            x = torch.cat(ups, dim=1)
        elif len(ups) == 1:
            x = ups[0]
        if len(self.deblocks) > len(self.blocks):
            x = self._run(self.deblocks[-1], x, nhwc)
        data_dict['spatial_features_2d'] = x
        return data_dict
