from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils


def _cfg(config, key, default=None):
    return config.get(key, default) if hasattr(config, 'get') else getattr(config, key, default)


def build_local_aggregation_module(input_channels, config):
    """(reference pointnet2_modules.py:10-27) only StackSAModuleMSG is built."""
    name = _cfg(config, 'NAME', 'StackSAModuleMSG')
    if name != 'StackSAModuleMSG':
        raise NotImplementedError(name)
    mlps = [[input_channels] + list(m) for m in _cfg(config, 'MLPS')]
    layer = StackSAModuleMSG(radii=_cfg(config, 'POOL_RADIUS'), nsamples=_cfg(config, 'NSAMPLE'), mlps=mlps, use_xyz=True, pool_method='max_pool')
    return layer, sum(m[-1] for m in mlps)


class StackSAModuleMSG(nn.Module):
    """Multi-scale set abstraction over stacked batches: ball query + group (HIP) -> shared 1x1 conv / BN / ReLU -> max over
    the neighbours. Same constructor, submodule names (groupers, mlps) and forward signature as the reference
    (ops/pointnet2/pointnet2_stack/pointnet2_modules.py:30-112)."""

    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            spec = list(spec)
            if use_xyz:
                spec[0] += 3
            layers = []
            for k in range(len(spec) - 1):
                layers += [nn.Conv2d(spec[k], spec[k + 1], kernel_size=1, bias=False), nn.BatchNorm2d(spec[k + 1]), nn.ReLU()]
            self.mlps.append(nn.Sequential(*layers))
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True):
        outs = []
        for k in range(len(self.groupers)):
            grouped, _ = self.groupers[k](xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)    # (M, C, nsample)
            x = self.mlps[k](grouped.permute(1, 0, 2).unsqueeze(dim=0))                               # (1, C', M, nsample)
            if self.pool_method == 'max_pool':
                x = F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(dim=-1)
            elif self.pool_method == 'avg_pool':
                x = F.avg_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(dim=-1)
            else:
                raise NotImplementedError
            outs.append(x.squeeze(dim=0).permute(1, 0))                                               # (M, C')
        return new_xyz, torch.cat(outs, dim=1)
