from collections import OrderedDict

import os

import torch.nn as nn

FUSE_CONV_BN = os.environ.get("SEEVCN_FUSE_CONV_BN", "1") != "0"      # 0: conv and BatchNorm as two autograd nodes (A/B runs, tests)


class SparseModule(nn.Module):
    """Marker base: modules that take and return a SparseConvTensor (spconv.SparseModule)."""


def _is_sparse(m):
    return isinstance(m, SparseModule)


def _hooked(*mods):
    """A module with forward (pre-)hooks must run through its own __call__: the fused paths below call kernels directly."""
    return any(m._forward_hooks or m._forward_pre_hooks for m in mods)


class SparseSequential(SparseModule):
    """Sequential that feeds SparseModules the sparse tensor and plain nn.Modules (BatchNorm1d, ReLU) the
    `.features` matrix — the behaviour post_act_block relies on (spconv_backbone.py:21-25)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError(f"index {idx} is out of range")
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        from .core import SparseConvTensor
        from . import norm
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if (FUSE_CONV_BN and i + 1 < len(mods) and hasattr(module, "fusable_with") and isinstance(input, SparseConvTensor)
                    and input.indices.shape[0] != 0 and module.fusable_with(mods[i + 1], input)
                    and not _hooked(module, mods[i + 1], *mods[i + 2:i + 3])):
                # conv -> BatchNorm1d [-> ReLU] as one autograd node (same kernels, half the host-side bookkeeping)
                relu = i + 2 < len(mods) and type(mods[i + 2]) is nn.ReLU
                input = module.forward_bn_relu(input, mods[i + 1], relu)
                i += 3 if relu else 2
                continue
            if _is_sparse(module):
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    if norm.fusable(module, input.features) and not _hooked(module, *mods[i + 1:i + 2]):
                        # BatchNorm1d [-> ReLU] in one pass over the features (sv_batchnorm_relu_forward)
                        relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                        input = input.replace_feature(norm.batch_norm_relu(module, input.features, relu))
                        i += 1 if relu else 0
                    else:
                        input = input.replace_feature(module(input.features))
            else:
                input = module(input)
            i += 1
        return input
