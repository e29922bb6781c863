"""Deterministic DynamicPillarVFE inputs shared by the golden generator and the test (PointPillars geometry, 2 scenes)."""
import numpy as np

RANGE = [0, -39.68, -3, 69.12, 39.68, 1]
VOXEL = [0.16, 0.16, 4]
GRID = [432, 496, 1]
CFG = dict(NAME='DynPillarVFE', WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, USE_NORM=True, NUM_FILTERS=[64, 64])


def make_points():
    import seevcn_amd.synth as synth
    rows = []
    for b in range(2):
        pts, _ = synth.make_scene(2000 + b, n_az=90)
        rng = np.random.default_rng(50 + b)
        inten = rng.uniform(size=(len(pts), 1)).astype(np.float32)
        extra = rng.uniform([-5, -45, -6], [75, 45, 5], (200, 3)).astype(np.float32)        # out-of-range in x/y, and z outside [-3,1] (kept!)
        xyz = np.concatenate([pts[:, :3], extra])
        inten = np.concatenate([inten, rng.uniform(size=(200, 1)).astype(np.float32)])
        rows.append(np.concatenate([np.full((len(xyz), 1), b, np.float32), xyz, inten], 1))
    out = np.concatenate(rows)
    return out[np.random.default_rng(9).permutation(len(out))].astype(np.float32)
