"""Oracle: dynamic voxelisation + mean VFE (numpy).  Test infrastructure only."""
import numpy as np


def dynamic_mean_vfe(points, pc_range, voxel_size, grid_size):
    """Follows DynamicMeanVFE.forward, detector3d/pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:38-76.

    points (P, 1+C) float32 [b,x,y,z,...] -> (voxel_features (V,C) f32, voxel_coords (V,4) i32 [b,z,y,x],
    point_to_voxel (P,) i32 with -1 for masked points).  Keys are int64 here (the reference's int32 keys
    agree while B*X*Y*Z < 2**31, :57-60).
    """
    points = np.asarray(points, np.float32)
    lo = np.asarray(pc_range[:3], np.float32)
    vs = np.asarray(voxel_size, np.float32)
    grid = np.asarray(grid_size, np.int64)
    # :53  floor((xyz - min) / voxel).int() in fp32
    f = np.floor((points[:, 1:4] - lo) / vs)
    mask = ((f >= 0) & (f < grid.astype(np.float32))).all(axis=1)           # :54
    pc = f[mask].astype(np.int64)
    pts = points[mask]
    sxyz, syz, sz = grid[0] * grid[1] * grid[2], grid[1] * grid[2], grid[2]  # :30-32
    key = pts[:, 0].astype(np.int64) * sxyz + pc[:, 0] * syz + pc[:, 1] * sz + pc[:, 2]  # :57-60
    unq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)  # :63
    data = pts[:, 1:]
    acc = np.zeros((len(unq), data.shape[1]), np.float32)
    np.add.at(acc, inv, data)                                                # scatter_mean :65 (sum ...
    feats = acc / cnt[:, None].astype(np.float32)                            #  ... / count)
    coords = np.stack([unq // sxyz, (unq % sxyz) // syz, (unq % syz) // sz, unq % sz], axis=1)  # :67-71
    coords = coords[:, [0, 3, 2, 1]].astype(np.int32)                        # :72
    p2v = np.full(len(points), -1, np.int32)
    p2v[mask] = inv.astype(np.int32)
    return feats, coords, p2v


def mean_vfe(voxels, num_points):
    """MeanVFE.forward, backbones_3d/vfe/mean_vfe.py:25-29."""
    s = np.asarray(voxels, np.float32).sum(axis=1, dtype=np.float32)
    d = np.maximum(np.asarray(num_points).reshape(-1, 1).astype(np.float32), 1.0)
    return s / d
