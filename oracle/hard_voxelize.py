"""Oracle: hard voxelisation + PillarVFE decoration (numpy / python loops).  Test infrastructure only.

The voxeliser itself is spconv's (`VoxelGeneratorV2` / `Point2VoxelCPU3d`, un-vendored third party — PARITY UNPINNED, see
oracle/spconv.py); this restates its published sequential algorithm as called from
detector3d/pcdet/datasets/processor/data_processor.py:115-143: points are visited in order; a new cell opens the next voxel
unless max_voxels are already open (the point is then skipped); a point is appended to its voxel unless it already holds
max_points points.  Coordinates are stored (z, y, x).  The decoration follows PillarVFE.forward (the reference's own code),
pinned by tests/golden/pillar_vfe.npz."""
import numpy as np

F = np.float32


def points_to_voxel(points, voxel_size, pc_range, max_points, max_voxels):
    points = np.asarray(points, F)
    vs, lo = np.asarray(voxel_size, F), np.asarray(pc_range[:3], F)
    grid = np.round((np.asarray(pc_range[3:6], np.float64) - np.asarray(pc_range[:3], np.float64)) / np.asarray(voxel_size, np.float64)).astype(np.int64)
    c = np.floor((points[:, :3] - lo) / vs)
    ok = ((c >= 0) & (c < grid.astype(F))).all(1)
    ci = c.astype(np.int64)
    voxels = np.zeros((max_voxels, max_points, points.shape[1]), F)
    coords = np.zeros((max_voxels, 3), np.int32)
    nump = np.zeros((max_voxels,), np.int32)
    table = {}
    nv = 0
    for i in range(len(points)):
        if not ok[i]:
            continue
        key = (int(ci[i, 2]), int(ci[i, 1]), int(ci[i, 0]))
        v = table.get(key, -1)
        if v == -1:
            if nv >= max_voxels:
                continue
            v = nv
            nv += 1
            table[key] = v
            coords[v] = key
        if nump[v] < max_points:
            voxels[v, nump[v]] = points[i]
            nump[v] += 1
    return voxels[:nv], coords[:nv], nump[:nv]


def pillar_decorate(voxels, nump, coords, voxel_size, pc_range, use_abs_xyz=True, with_distance=False):
    """PillarVFE.forward, backbones_3d/vfe/pillar_vfe.py:94-118. coords (V,4) [b,z,y,x]."""
    v = np.asarray(voxels, F)
    n = np.asarray(nump).astype(F).reshape(-1, 1, 1)
    mean = v[:, :, :3].sum(1, keepdims=True, dtype=F) / n
    f_cluster = v[:, :, :3] - mean
    off = [voxel_size[i] / 2 + pc_range[i] for i in range(3)]
    f_center = np.stack([v[:, :, 0] - (coords[:, 3].astype(F)[:, None] * F(voxel_size[0]) + F(off[0])),
                         v[:, :, 1] - (coords[:, 2].astype(F)[:, None] * F(voxel_size[1]) + F(off[1])),
                         v[:, :, 2] - (coords[:, 1].astype(F)[:, None] * F(voxel_size[2]) + F(off[2]))], -1)
    feats = [v if use_abs_xyz else v[..., 3:], f_cluster, f_center]
    if with_distance:
        feats.append(np.linalg.norm(v[:, :, :3], axis=2, keepdims=True).astype(F))
    f = np.concatenate(feats, -1)
    mask = (np.arange(v.shape[1])[None, :] < np.asarray(nump)[:, None]).astype(F)[..., None]
    return (f * mask).astype(F)
