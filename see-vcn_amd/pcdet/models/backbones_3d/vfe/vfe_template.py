"""Base class of the voxel feature encoders; contract of the reference's VFETemplate (backbones_3d/vfe/vfe_template.py:4-22):
constructed with `model_cfg` plus keyword geometry, reports its channel count, maps batch_dict -> batch_dict."""
import abc

import torch.nn as nn


class VFETemplate(nn.Module, metaclass=abc.ABCMeta):
    def __init__(self, model_cfg, **_geometry):
        nn.Module.__init__(self)
        self.model_cfg = model_cfg

    @abc.abstractmethod
    def get_output_feature_dim(self):
        """number of channels of `voxel_features` / `pillar_features`"""

    @abc.abstractmethod
    def forward(self, batch_dict, **kwargs):
        """adds `voxel_features` (or `pillar_features`) to batch_dict and returns it"""
