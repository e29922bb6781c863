"""Generate tests/golden/dyn_voxel_*.npz by running the REFERENCE's DynamicMeanVFE
(detector3d/pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:38-76) and MeanVFE (mean_vfe.py:14-31) on CPU.

Run only in the build container (needs /root/reference):  python tests/golden/make_voxel_golden.py
`torch_scatter.scatter_mean` is absent here; _refimport provides a functional stand-in (index_add / count),
so the *indices* are the reference's own arithmetic and the means are sum/count in fp32.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refimport as R  # noqa: E402

R.import_pcdet()
from pcdet.models.backbones_3d.vfe.dynamic_mean_vfe import DynamicMeanVFE  # noqa: E402
from pcdet.models.backbones_3d.vfe.mean_vfe import MeanVFE  # noqa: E402
import seevcn_amd.synth as synth  # noqa: E402

CASES = {
    # name: (range, voxel_size, scene kwargs, batch)
    "kitti": ([0, -40, -3, 70.4, 40, 1], [0.05, 0.05, 0.1], dict(n_az=120), 2),
    "da": ([-75.2, -75.2, -2, 75.2, 75.2, 4], [0.1, 0.1, 0.15], dict(n_az=90, az=(-180.0, 180.0), z_shift=1.8), 3),
    "coarse": ([0, -39.68, -3, 69.12, 39.68, 1], [0.64, 0.64, 0.8], dict(n_az=60), 2),
}

for name, (pc_range, vsize, kw, bs) in CASES.items():
    points, _ = synth.make_scene_batch(bs, seed=2000, **kw)
    rng = np.random.default_rng(7)
    # add out-of-range points, points exactly on the range borders, and exact duplicates
    junk = rng.uniform(-100, 100, size=(200, 4)).astype(np.float32)
    junk[:, 0] = rng.integers(0, bs, size=200)
    border = np.array([[0, pc_range[0], pc_range[1], pc_range[2]],
                       [0, pc_range[3], pc_range[4], pc_range[5]],
                       [bs - 1, pc_range[0] + vsize[0], pc_range[1] + vsize[1], pc_range[2] + vsize[2]],
                       [bs - 1, np.nextafter(np.float32(pc_range[3]), np.float32(-1e9)), 0.0, 0.0]], np.float32)
    points = np.concatenate([points, junk, border, points[:50]], axis=0)
    points = points[rng.permutation(len(points))]
    grid = np.round((np.array(pc_range[3:6]) - np.array(pc_range[0:3])) / np.array(vsize)).astype(np.int64)
    vfe = DynamicMeanVFE(model_cfg={}, num_point_features=3, voxel_size=vsize, grid_size=grid.tolist(),
                         point_cloud_range=pc_range)
    bd = vfe({"batch_size": bs, "points": torch.from_numpy(points)})
    out = os.path.join(HERE, f"dyn_voxel_{name}.npz")
    np.savez_compressed(out, points=points, pc_range=np.array(pc_range, np.float32),
                        voxel_size=np.array(vsize, np.float32), grid_size=grid.astype(np.int32),
                        batch_size=np.int32(bs),
                        voxel_coords=bd["voxel_coords"].numpy().astype(np.int32),
                        voxel_features=bd["voxel_features"].numpy())
    print(name, points.shape, "->", bd["voxel_coords"].shape, os.path.getsize(out))

# MeanVFE: hard-voxel layout (V, max_pts, C) + counts
rng = np.random.default_rng(11)
voxels = rng.normal(size=(500, 5, 3)).astype(np.float32)
nump = rng.integers(0, 6, size=500).astype(np.int32)  # includes 0 (clamp_min) and 5
for i, n in enumerate(nump):
    voxels[i, n:] = 0
bd = MeanVFE(model_cfg={}, num_point_features=3)(
    {"voxels": torch.from_numpy(voxels), "voxel_num_points": torch.from_numpy(nump)})
np.savez_compressed(os.path.join(HERE, "mean_vfe.npz"), voxels=voxels, voxel_num_points=nump,
                    voxel_features=bd["voxel_features"].numpy())
print("mean_vfe ok")
