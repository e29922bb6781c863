"""Dynamic voxelisation / mean VFE: oracle vs the reference's goldens (CPU), HIP vs both (GPU)."""
import os

import numpy as np
import pytest

from oracle import voxelize as ov

CASES = ["kitti", "da", "coarse"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"dyn_voxel_{name}.npz"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(golden_dir, name):
    g = _load(golden_dir, name)
    feats, coords, p2v = ov.dynamic_mean_vfe(g["points"], g["pc_range"], g["voxel_size"], g["grid_size"])
    assert np.array_equal(coords, g["voxel_coords"])          # bit-exact indices, same order
    np.testing.assert_allclose(feats, g["voxel_features"], rtol=1e-6, atol=1e-6)
    assert (p2v >= 0).sum() == (p2v != -1).sum() and p2v.max() == len(coords) - 1


def test_oracle_mean_vfe_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "mean_vfe.npz"))
    np.testing.assert_allclose(ov.mean_vfe(g["voxels"], g["voxel_num_points"]), g["voxel_features"], rtol=1e-6, atol=1e-6)


def test_oracle_empty_and_all_masked():
    f, c, p = ov.dynamic_mean_vfe(np.zeros((0, 4), np.float32), [0, 0, 0, 1, 1, 1], [0.5, 0.5, 0.5], [2, 2, 2])
    assert f.shape == (0, 3) and c.shape == (0, 4)
    pts = np.array([[0, 5, 5, 5], [0, -1, 0, 0]], np.float32)
    f, c, p = ov.dynamic_mean_vfe(pts, [0, 0, 0, 1, 1, 1], [0.5, 0.5, 0.5], [2, 2, 2])
    assert len(c) == 0 and (p == -1).all()


# ------------------------------------------------------------------------------------------ GPU
def _run_hip(points, pc_range, voxel_size, grid, bs, cuda, **kw):
    import torch
    from seevcn_amd.pcdet.ops import voxel_ops
    out = voxel_ops.voxelize_dynamic(torch.from_numpy(points).to(cuda), pc_range, voxel_size, grid, bs,
                                     return_point_to_voxel=True, **kw)
    torch.cuda.synchronize()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference_golden(golden_dir, cuda, hip_lib, name):
    g = _load(golden_dir, name)
    feats, coords, p2v = _run_hip(g["points"], g["pc_range"], g["voxel_size"], g["grid_size"], int(g["batch_size"]), cuda)
    assert np.array_equal(coords.cpu().numpy(), g["voxel_coords"])     # bit-exact, ascending-key order
    np.testing.assert_allclose(feats.cpu().numpy(), g["voxel_features"], rtol=1e-5, atol=1e-5)
    _, _, p2v_o = ov.dynamic_mean_vfe(g["points"], g["pc_range"], g["voxel_size"], g["grid_size"])
    assert np.array_equal(p2v.cpu().numpy(), p2v_o)
    # second call on the same (cleaned) persistent workspace must give the same answer
    feats2, coords2, _ = _run_hip(g["points"], g["pc_range"], g["voxel_size"], g["grid_size"], int(g["batch_size"]), cuda)
    assert np.array_equal(coords2.cpu().numpy(), g["voxel_coords"])


@pytest.mark.gpu
def test_hip_module_dropin(golden_dir, cuda, hip_lib):
    """Through the registry name the reference's config uses (VFE.NAME: DynMeanVFE)."""
    import torch
    from seevcn_amd.pcdet.models.backbones_3d import vfe
    g = _load(golden_dir, "da")
    m = vfe.__all__["DynMeanVFE"](model_cfg={}, num_point_features=3, voxel_size=g["voxel_size"].tolist(),
                                  grid_size=g["grid_size"].tolist(), point_cloud_range=g["pc_range"].tolist())
    bd = m({"batch_size": int(g["batch_size"]), "points": torch.from_numpy(g["points"]).to(cuda)})
    assert np.array_equal(bd["voxel_coords"].cpu().numpy(), g["voxel_coords"])
    np.testing.assert_allclose(bd["voxel_features"].cpu().numpy(), g["voxel_features"], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_hip_edge_cases(cuda, hip_lib):
    rng, vs, grid = [0, 0, 0, 1, 1, 1], [0.5, 0.5, 0.5], [2, 2, 2]
    # empty input
    f, c, p = _run_hip(np.zeros((0, 4), np.float32), rng, vs, grid, 1, cuda)
    assert f.shape == (0, 3) and c.shape == (0, 4)
    # everything masked, NaN/inf coordinates, bad batch index
    pts = np.array([[0, 5, 5, 5], [0, -1, 0, 0], [0, np.nan, 0, 0], [0, np.inf, 0, 0], [7, 0.1, 0.1, 0.1]], np.float32)
    f, c, p = _run_hip(pts, rng, vs, grid, 1, cuda)
    assert len(c) == 0 and (p.cpu().numpy() == -1).all()
    # all points in one voxel (maximum collisions) + 5 feature columns
    r = np.random.default_rng(0)
    pts = np.concatenate([np.zeros((5000, 1), np.float32), r.uniform(0.5, 0.999, (5000, 3)).astype(np.float32),
                          r.normal(size=(5000, 2)).astype(np.float32)], axis=1)
    f, c, p = _run_hip(pts, rng, vs, grid, 1, cuda)
    assert c.cpu().numpy().tolist() == [[0, 1, 1, 1]]
    np.testing.assert_allclose(f.cpu().numpy()[0], pts[:, 1:].mean(0), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_hip_full_size_properties(cuda, hip_lib):
    """BASELINE config-3 size (16 scenes x ~17-20k pts, KITTI grid): oracle equality + size-independent properties."""
    import seevcn_amd.synth as synth
    pts, _ = synth.make_scene_batch(16, seed=2000)
    pc_range, vs = [0, -40, -3, 70.4, 40, 1], [0.05, 0.05, 0.1]
    grid = [1408, 1600, 40]
    f, c, p = _run_hip(pts, pc_range, vs, grid, 16, cuda)
    fo, co, po = ov.dynamic_mean_vfe(pts, pc_range, vs, grid)
    c, p, f = c.cpu().numpy(), p.cpu().numpy(), f.cpu().numpy()
    assert np.array_equal(c, co) and np.array_equal(p, po)
    np.testing.assert_allclose(f, fo, rtol=1e-5, atol=1e-5)
    # sortedness by the reference key, uniqueness, count conservation
    key = ((c[:, 0].astype(np.int64) * grid[0] + c[:, 3]) * grid[1] + c[:, 2]) * grid[2] + c[:, 1]
    assert (np.diff(key) > 0).all()
    assert np.bincount(p[p >= 0], minlength=len(c)).sum() == (p >= 0).sum()
    # permutation invariance of the voxel set (idempotence of the index clean-up as well)
    perm = np.random.default_rng(1).permutation(len(pts))
    f2, c2, _ = _run_hip(pts[perm], pc_range, vs, grid, 16, cuda)
    assert np.array_equal(c2.cpu().numpy(), c)
    np.testing.assert_allclose(f2.cpu().numpy(), f, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_hip_mean_vfe(golden_dir, cuda, hip_lib):
    import torch
    from seevcn_amd.pcdet.models.backbones_3d import vfe
    g = np.load(os.path.join(golden_dir, "mean_vfe.npz"))
    m = vfe.__all__["MeanVFE"](model_cfg={}, num_point_features=3)
    bd = m({"voxels": torch.from_numpy(g["voxels"]).to(cuda), "voxel_num_points": torch.from_numpy(g["voxel_num_points"]).to(cuda)})
    np.testing.assert_allclose(bd["voxel_features"].cpu().numpy(), g["voxel_features"], rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_hip_alternating_grids_keep_their_indices_clean(golden_dir, cuda, hip_lib):
    """Calls on different grids in one process (each grid size owns a persistent index): results stay bit-exact in any order."""
    names = list(CASES)
    for name in names + names[::-1] + names:
        g = _load(golden_dir, name)
        _, coords, _ = _run_hip(g["points"], g["pc_range"], g["voxel_size"], g["grid_size"], int(g["batch_size"]), cuda)
        assert np.array_equal(coords.cpu().numpy(), g["voxel_coords"]), name
