"""Deterministic CenterHead inputs shared by the golden generator and the test."""
import numpy as np

RANGE = [-12.8, -12.8, -5.0, 12.8, 12.8, 3.0]
VOXEL = [0.1, 0.1, 0.2]
GRID = [256, 256, 40]


def make_inputs():
    rng = np.random.default_rng(31)
    feat = (rng.normal(size=(2, 24, 32, 32)) * 0.7).astype(np.float32)
    gt = np.zeros((2, 30, 10), np.float32)
    for b in range(2):
        n = 22 + 5 * b
        gt[b, :n, 0:2] = rng.uniform(-13.5, 13.5, (n, 2))          # a few centres fall outside the range (clamped by the reference)
        gt[b, :n, 2] = rng.uniform(-2, 0, n)
        gt[b, :n, 3:6] = rng.uniform([0.4, 0.4, 0.8], [7.0, 2.8, 3.0], (n, 3))
        gt[b, :n, 6] = rng.uniform(-3.2, 3.2, n)
        gt[b, :n, 7:9] = rng.normal(0, 2, (n, 2))
        gt[b, :n, 9] = rng.integers(1, 11, n)
    gt[0, 3, 3] = 0.0                                                # degenerate box: skipped (dx <= 0)
    return {'feat': feat, 'gt_boxes': gt}
