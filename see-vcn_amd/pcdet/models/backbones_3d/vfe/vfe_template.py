import torch.nn as nn


class VFETemplate(nn.Module):
    """Same base contract as the reference's VFETemplate (backbones_3d/vfe/vfe_template.py:4-22)."""

    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg

    def get_output_feature_dim(self):
        raise NotImplementedError

    def forward(self, **kwargs):
        raise NotImplementedError
