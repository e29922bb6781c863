import torch.nn as nn

from .....spconv import functional as Fsp
from ....utils.common_utils import cfg_get


class PointPillarScatter(nn.Module):
    """Drop-in for the reference PointPillarScatter (backbones_2d/map_to_bev/pointpillar_scatter.py:5-37): pillar features
    (V,C) + coords [b,z,y,x] -> (B, C*nz, ny, nx) with the single-pass HIP scatter (no per-scene python loop, no memset)."""

    def __init__(self, model_cfg, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = cfg_get(model_cfg, 'NUM_BEV_FEATURES')
        self.nx, self.ny, self.nz = (int(g) for g in grid_size)
        assert self.nz == 1

    def forward(self, batch_dict, **kwargs):
        pillar_features, coords = batch_dict['pillar_features'], batch_dict['voxel_coords']
        batch_size = batch_dict['batch_size'] if 'batch_size' in batch_dict else int(coords[:, 0].max().item()) + 1
        dense = Fsp.sparse_to_dense(pillar_features, coords.int(), batch_size, [self.nz, self.ny, self.nx])   # (B,C,nz,ny,nx)
        batch_dict['spatial_features'] = dense.view(batch_size, self.num_bev_features * self.nz, self.ny, self.nx)
        return batch_dict
