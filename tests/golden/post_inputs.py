"""Deterministic inputs for the VCN post-processing tests: (partial, coarse) pairs shaped like VCN.inference's
in_pc / output (B,1024,3) float32 -- coarse = noisy samples of a car-sized box surface, partial = a view-dependent subset
tiled to 1024 the way ResamplePoints does (np.tile + permutation pick)."""
import numpy as np


def _box_surface(rng, n, dims):
    face = rng.integers(0, 6, n)
    uvw = rng.uniform(-0.5, 0.5, (n, 3))
    ax = face // 2
    uvw[np.arange(n), ax] = np.where(face % 2 == 0, -0.5, 0.5)
    return uvw * np.asarray(dims)


def make_pairs(n_objects=8, seed=77, n_points=1024):
    rng = np.random.default_rng(seed)
    partial = np.zeros((n_objects, n_points, 3), np.float32)
    coarse = np.zeros((n_objects, n_points, 3), np.float32)
    sizes = [30, 57, 120, 260, 400, 9, 1024, 0]
    for b in range(n_objects):
        dims = np.array([3.9, 1.6, 1.56]) * rng.uniform(0.8, 1.2)
        c = _box_surface(rng, n_points, dims) + rng.normal(0, 0.02, (n_points, 3))
        if b % 3 == 2:                                   # detached blob: a second, smaller cluster for DBSCAN
            c[:60] = rng.normal(0, 0.05, (60, 3)) + np.array([0.0, 3.0, 0.0])
        coarse[b] = c.astype(np.float32)
        ni = sizes[b % len(sizes)]
        if ni == 0:
            continue                                     # zero-padded object (VCN.py:55-59 pads the last chunk with zeros)
        surf = _box_surface(rng, 4000, dims)
        vis = surf[(surf[:, 0] < -0.3 * dims[0]) | (surf[:, 1] < -0.3 * dims[1])]
        pts = (vis[rng.permutation(len(vis))[:ni]] + rng.normal(0, 0.02, (ni, 3))).astype(np.float32)
        if b % 3 == 2:
            pts[:3] = coarse[b, :3]
        reps = int(np.ceil(n_points / ni))
        tiled = np.tile(pts, (reps, 1))
        partial[b] = tiled[rng.permutation(len(tiled))[:n_points]]
    return partial, coarse
