from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..... import _lib
from . import pointnet2_stack_cuda, pointnet2_utils


import os

FUSED_SA_OFF = os.environ.get("SEEVCN_FUSED_SA", "1") == "0"      # 0: always the torch path (A/B runs, tests)
TRAIN_SA_OFF = os.environ.get("SEEVCN_TRAIN_SA", "1") == "0"      # 0: training goes through the (M, C, nsample) Conv2d path of the reference


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

    def train(self, mode=True):
        self._fused_weights = {}              # folded BatchNorm: re-made after any switch of mode (fused optimisers do not bump tensor versions)
        return super().train(mode)

    def _fused_ok(self, k, xyz, new_xyz, features):
        """The fused eval kernel (sv_sa_mlp_max) takes this scale: 2 conv-BN-ReLU layers of 16..64 channels, 16 or 32 neighbours, max pooling,
        feature channels a multiple of 16 up to 128 (or none), no gradient needed."""
        if self.training or self.pool_method != 'max_pool' or FUSED_SA_OFF or not (xyz.is_cuda and new_xyz.is_cuda):
            return False
        if torch.is_grad_enabled() and ((features is not None and features.requires_grad) or any(p.requires_grad for p in self.mlps[k].parameters())):
            return False
        layers = list(self.mlps[k])
        if len(layers) != 6 or not all(isinstance(layers[i], nn.Conv2d) and isinstance(layers[i + 1], nn.BatchNorm2d) for i in (0, 3)):
            return False
        c = 0 if features is None else features.shape[1]
        c1, c2 = layers[0].out_channels, layers[3].out_channels
        ns = self.groupers[k].nsample
        return (c % 16 == 0 and c <= 128 and layers[0].in_channels == c + 3 and ns in (16, 32) and all(v % 16 == 0 and 16 <= v <= 64 for v in (c1, c2))
                and layers[0].bias is None and layers[3].bias is None and (features is None or features.dtype == torch.float32))

    def _prepared(self, k, device):
        hit = self._fused_weights.get(k) if hasattr(self, '_fused_weights') else None
        if hit is not None:
            return hit
        lib = _lib.load()
        layers = list(self.mlps[k])
        out = []
        for conv, bn, first in ((layers[0], layers[1], 1), (layers[3], layers[4], 0)):
            co, ci = conv.out_channels, conv.in_channels
            kp = 16 * ((ci - 3) // 16 + 1) if first else ci
            w = torch.empty((co, kp), dtype=torch.float32, device=device)
            b = torch.empty((co,), dtype=torch.float32, device=device)
            _lib.check(lib.sv_sa_prepare_weights(_lib.ptr(conv.weight.detach().reshape(co, ci).contiguous()), _lib.ptr(bn.weight.detach()), _lib.ptr(bn.bias.detach()),
                                                 _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), float(bn.eps), co, ci, first, _lib.ptr(w), _lib.ptr(b),
                                                 _lib.stream()), "sv_sa_prepare_weights")
            out += [w, b]
        if not hasattr(self, '_fused_weights'):
            self._fused_weights = {}
        self._fused_weights[k] = tuple(out)
        return self._fused_weights[k]

    def _fused_scale(self, k, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features):
        """ball query (HIP) -> sv_sa_mlp_max: gather + both MLP layers on the matrix core + max over the neighbours in ONE launch; no
        (M, C+3, nsample) tensor (the torch path below materialises it and runs Conv2d / BatchNorm2d / max_pool2d over it)."""
        lib = _lib.load()
        g = self.groupers[k]
        M, B = new_xyz.shape[0], xyz_batch_cnt.shape[0]
        idx = torch.zeros((M, g.nsample), dtype=torch.int32, device=new_xyz.device)
        pointnet2_stack_cuda.ball_query_wrapper(B, M, g.radius, g.nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
        row_start = pointnet2_stack_cuda._row_start(new_xyz_batch_cnt, xyz_batch_cnt, M)
        w1, b1, w2, b2 = self._prepared(k, new_xyz.device)
        c = 0 if features is None else features.shape[1]
        out = torch.empty((M, w2.shape[0]), dtype=torch.float32, device=new_xyz.device)
        _lib.check(lib.sv_sa_mlp_max(_lib.ptr(xyz), _lib.ptr(features) if c else None, _lib.ptr(new_xyz), _lib.ptr(idx), _lib.ptr(row_start), M, c, g.nsample,
                                     _lib.ptr(w1), _lib.ptr(b1), w1.shape[0], _lib.ptr(w2), _lib.ptr(b2), w2.shape[0], _lib.ptr(out), _lib.stream()),
                   "sv_sa_mlp_max")
        return out

    def _train_ok(self, k, xyz, new_xyz, features):
        """The training kernels (sv_sa_train_forward / _backward) take this scale: two conv-BN-ReLU layers of 16..64 channels without conv bias,
        16 or 32 neighbours, max pooling, feature channels a multiple of 16 up to 128 (or none), fp32 on the GPU, xyz in the features,
        BatchNorm2d in training mode with affine parameters and running statistics."""
        if (TRAIN_SA_OFF or not self.training or self.pool_method != 'max_pool' or not (xyz.is_cuda and new_xyz.is_cuda) or not self.groupers[k].use_xyz
                or new_xyz.shape[0] * self.groupers[k].nsample < 2):
            return False
        layers = list(self.mlps[k])
        if len(layers) != 6 or (features is not None and features.dtype != torch.float32):
            return False
        for i in (0, 3):
            conv, bn = layers[i], layers[i + 1]
            if not (isinstance(conv, nn.Conv2d) and type(bn) is nn.BatchNorm2d and isinstance(layers[i + 2], nn.ReLU) and conv.bias is None
                    and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None):
                return False
            if conv._forward_hooks or conv._forward_pre_hooks or bn._forward_hooks or bn._forward_pre_hooks:
                return False
        c = 0 if features is None else features.shape[1]
        c1, c2 = layers[0].out_channels, layers[3].out_channels
        return (c % 16 == 0 and c <= 128 and layers[0].in_channels == c + 3 and layers[3].in_channels == c1 and self.groupers[k].nsample in (16, 32)
                and all(v % 16 == 0 and 16 <= v <= 64 for v in (c1, c2)) and layers[1].eps == layers[4].eps and layers[1].momentum == layers[4].momentum)

    def _train_scale(self, k, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features):
        """ball query (HIP) -> SAScaleTrain: the whole scale (gather, both MLP layers with batch-statistics BatchNorm, max over the neighbours) and its
        backward on the hand-written MFMA kernels; same parameters, running statistics and num_batches_tracked as the Conv2d / BatchNorm2d path."""
        g = self.groupers[k]
        M, B = new_xyz.shape[0], xyz_batch_cnt.shape[0]
        idx = torch.zeros((M, g.nsample), dtype=torch.int32, device=new_xyz.device)
        pointnet2_stack_cuda.ball_query_wrapper(B, M, g.radius, g.nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
        row_start = pointnet2_stack_cuda._row_start(new_xyz_batch_cnt, xyz_batch_cnt, M)
        conv1, bn1, _, conv2, bn2, _ = list(self.mlps[k])
        return pointnet2_utils.sa_scale_train(xyz, features, new_xyz, idx, row_start, conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias,
                                              bn1.running_mean, bn1.running_var, bn1.num_batches_tracked, bn2.running_mean, bn2.running_var,
                                              bn2.num_batches_tracked, bn1.momentum, bn1.eps)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True):
        outs = []
        xyz, new_xyz = xyz.contiguous(), new_xyz.contiguous()
        feats_c = features.contiguous() if features is not None else None
        for k in range(len(self.groupers)):
            if self._fused_ok(k, xyz, new_xyz, feats_c):
                outs.append(self._fused_scale(k, xyz, xyz_batch_cnt.contiguous(), new_xyz, new_xyz_batch_cnt.contiguous(), feats_c))
                continue
            if self._train_ok(k, xyz, new_xyz, feats_c):
                outs.append(self._train_scale(k, xyz, xyz_batch_cnt.contiguous(), new_xyz, new_xyz_batch_cnt.contiguous(), feats_c))
                continue
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
