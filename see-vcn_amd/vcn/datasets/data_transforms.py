import numpy as np


class ResamplePoints(object):
    """Drop or duplicate points so that a cloud has exactly n points: tile up, then keep the first n of a random
    permutation drawn from numpy's global RNG (same contract and RNG consumption as the reference,
    see/surface_completion/models/vcn/datasets/data_transforms.py:247-262)."""

    def __init__(self, parameters):
        self.n_points = parameters['n_points']

    def __call__(self, pts):
        tiled = np.tile(pts, (int(np.ceil(self.n_points / len(pts))), 1))
        choice = np.random.permutation(tiled.shape[0])
        return tiled[choice[:self.n_points]]
