import torch

from . import functional as Fsp


class SparseConvTensor:
    """features (N,C) fp32, indices (N,4) int32 [b,z,y,x], spatial_shape [Z,Y,X], batch_size.

    Same constructor and attributes as spconv's SparseConvTensor as the reference uses it
    (detector3d/pcdet/models/backbones_3d/spconv_backbone.py:141-146); `indice_dict` caches rulebooks by
    `indice_key` so that e.g. both 'subm1' convolutions share one table (spconv_backbone.py:78,85)."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, indice_dict=None):
        self._features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}
        self.grid = grid

    @property
    def features(self):
        return self._features

    @features.setter
    def features(self, value):  # spconv 1.x style assignment (pcdet/utils/spconv_utils.py:32-33)
        self._features = value

    def replace_feature(self, new_features):
        """spconv 2.x style (pcdet/utils/spconv_utils.py:28-31): new tensor sharing indices and rulebooks."""
        return SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, self.grid, self.indice_dict)

    @property
    def spatial_size(self):
        n = 1
        for s in self.spatial_shape:
            n *= s
        return n

    def find_indice_pair(self, key):
        return None if key is None else self.indice_dict.get(key)

    def dense(self, channels_first=True):
        """(B, C, D, H, W) (height_compression.py:21); channels_first=False gives (B, D, H, W, C)."""
        out = Fsp.sparse_to_dense(self._features, self.indices, self.batch_size, self.spatial_shape)
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / max(self.spatial_size * self.batch_size, 1)
