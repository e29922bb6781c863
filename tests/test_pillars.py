"""PointPillars path: hard voxelisation (first-come semantics), PillarVFE, PointPillarScatter."""
import os

import numpy as np
import pytest
import torch

from oracle import hard_voxelize as ohv
from seeding import seeded_state_dict
from tolerances import assert_close_per_channel
from seevcn_amd.pcdet import model_cfgs as C


def test_oracle_pillar_decorate_reproduces_reference_golden(golden_dir):
    """the golden's pillar_features came from the reference PillarVFE; re-derive them from the oracle decoration + the same
    seeded Linear/BN to pin oracle/hard_voxelize.pillar_decorate"""
    from seevcn_amd.pcdet.models.backbones_3d import vfe
    g = np.load(os.path.join(golden_dir, "pillar_vfe.npz"))
    m = vfe.__all__["PillarVFE"](model_cfg=C.PP_VFE, num_point_features=4, voxel_size=C.PP_VOXEL["VOXEL_SIZE"], point_cloud_range=C.PP_RANGE).eval()
    m.load_state_dict(seeded_state_dict(m, seed=5))
    f = ohv.pillar_decorate(g["voxels"], g["voxel_num_points"], g["voxel_coords"], C.PP_VOXEL["VOXEL_SIZE"], C.PP_RANGE)
    with torch.no_grad():
        out = m.pfn_layers[0](torch.from_numpy(f)).squeeze().numpy()
    np.testing.assert_allclose(out, g["pillar_features"], rtol=1e-4, atol=1e-5)


def test_oracle_hard_voxelize_caps_and_order():
    pts = np.array([[0.1, 0.1, 0.1, 1], [5.1, 0.1, 0.1, 2], [0.2, 0.2, 0.1, 3], [0.3, 0.1, 0.2, 4], [9.1, 9.1, 0.1, 5], [5.2, 0.1, 0.1, 6],
                    [-1, 0, 0, 7]], np.float32)
    v, c, n = ohv.points_to_voxel(pts, [1, 1, 1], [0, 0, 0, 10, 10, 1], max_points=2, max_voxels=2)
    assert c.tolist() == [[0, 0, 0], [0, 0, 5]] and n.tolist() == [2, 2]
    assert v[0, :, 3].tolist() == [1, 3] and v[1, :, 3].tolist() == [2, 6]      # 4th point dropped (voxel full), 5th dropped (voxel cap)


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_hard_voxelize_matches_oracle(cuda, hip_lib):
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet.ops import voxel_ops
    scenes, feats = [], []
    for b in range(3):
        p, _ = synth.make_scene(2000 + b, n_az=(120 if b < 2 else 8))
        p = np.concatenate([p, np.random.default_rng(b).uniform(size=(len(p), 1)).astype(np.float32)], 1)
        p = p[np.random.default_rng(b).permutation(len(p))]            # shuffled like DataProcessor.shuffle_points
        scenes.append(p)
    allp = torch.from_numpy(np.concatenate(scenes)).to(cuda)
    for (vs, rg, mp, mv) in [(C.PP_VOXEL["VOXEL_SIZE"], C.PP_RANGE, 32, 16000), ([0.05, 0.05, 0.1], C.KITTI_RANGE, 5, 16000),
                             ([0.4, 0.4, 0.8], C.KITTI_RANGE, 3, 700)]:
        grid = np.round((np.array(rg[3:]) - np.array(rg[:3])) / np.array(vs)).astype(np.int64)
        vox, crd, nmp, nv = voxel_ops.voxelize_hard(allp, 0, 4, [len(s) for s in scenes], rg, vs, grid, mp, mv)
        torch.cuda.synchronize()
        for b, s in enumerate(scenes):
            ov, oc, on = ohv.points_to_voxel(s, vs, rg, mp, mv)
            n = int(nv[b])
            assert n == len(oc), (b, n, len(oc))
            assert np.array_equal(crd[b, :n].cpu().numpy(), oc)            # same voxels in the same (first-appearance) order
            assert np.array_equal(nmp[b, :n].cpu().numpy(), on)
            assert np.array_equal(vox[b, :n].cpu().numpy(), ov)            # same points in the same slots, bit-exact copies
            assert float(vox[b, n:].abs().sum()) == 0.0


@pytest.mark.gpu
def test_hip_pillar_vfe_scatter_match_reference_golden(golden_dir, cuda, hip_lib):
    from seevcn_amd.pcdet.models.backbones_2d import map_to_bev
    from seevcn_amd.pcdet.models.backbones_3d import vfe
    g = np.load(os.path.join(golden_dir, "pillar_vfe.npz"))
    m = vfe.__all__["PillarVFE"](model_cfg=C.PP_VFE, num_point_features=4, voxel_size=C.PP_VOXEL["VOXEL_SIZE"], point_cloud_range=C.PP_RANGE)
    m.load_state_dict(seeded_state_dict(m, seed=5))
    m = m.to(cuda).eval()
    grid = np.round((np.array(C.PP_RANGE[3:]) - np.array(C.PP_RANGE[:3])) / np.array(C.PP_VOXEL["VOXEL_SIZE"])).astype(np.int64)
    sc = map_to_bev.__all__["PointPillarScatter"](model_cfg=C.PP_MAP_TO_BEV, grid_size=grid)
    with torch.no_grad():
        bd = m({"voxels": torch.from_numpy(g["voxels"]).to(cuda), "voxel_num_points": torch.from_numpy(g["voxel_num_points"]).to(cuda),
                "voxel_coords": torch.from_numpy(g["voxel_coords"]).to(cuda), "batch_size": 2})
        bd = sc(bd)
    np.testing.assert_allclose(bd["pillar_features"].cpu().numpy(), g["pillar_features"], rtol=1e-3, atol=1e-5)
    sf = bd["spatial_features"]
    assert list(sf.shape) == g["spatial_shape"].tolist() == [2, 64, 496, 432]
    assert int((sf != 0).sum()) == int(g["spatial_nnz"])
    assert abs(float(sf.double().sum()) - float(g["spatial_sum"])) < 1e-3 * abs(float(g["spatial_sum"]))
    # value placement: out[b, :, y, x] == pillar feature
    c = g["voxel_coords"]
    got = sf[torch.from_numpy(c[:, 0]).long(), :, torch.from_numpy(c[:, 2]).long(), torch.from_numpy(c[:, 3]).long()].cpu().numpy()
    np.testing.assert_allclose(got, g["pillar_features"], rtol=1e-3, atol=1e-5)


def test_dyn_pillar_vfe_state_dict_keys():
    from dynpillar_inputs import CFG, GRID, RANGE, VOXEL
    from seevcn_amd.pcdet.models.backbones_3d import vfe
    m = vfe.__all__['DynPillarVFE'](model_cfg=CFG, num_point_features=4, voxel_size=VOXEL, grid_size=GRID, point_cloud_range=RANGE)
    sd = m.state_dict()
    assert sd['pfn_layers.0.linear.weight'].shape == (32, 10) and sd['pfn_layers.1.linear.weight'].shape == (64, 64)
    assert 'pfn_layers.1.norm.running_var' in sd and m.get_output_feature_dim() == 64


@pytest.mark.gpu
def test_hip_dyn_pillar_vfe_matches_reference_golden(golden_dir, cuda, hip_lib):
    """DynamicPillarVFE (D3): pillar coordinates and order bit-exact, features / running stats / weight gradients within 1e-3
    of the reference's own module (tests/golden/make_dynpillar_golden.py), train mode (batch-statistics BatchNorm)."""
    from dynpillar_inputs import CFG, GRID, RANGE, VOXEL, make_points
    from seeding import seeded_state_dict
    from seevcn_amd.pcdet.models.backbones_3d import vfe
    g = np.load(os.path.join(golden_dir, "dyn_pillar_vfe.npz"))
    m = vfe.__all__['DynPillarVFE'](model_cfg=CFG, num_point_features=4, voxel_size=VOXEL, grid_size=GRID, point_cloud_range=RANGE)
    m.load_state_dict(seeded_state_dict(m, seed=11))
    m = m.to(cuda).train()
    bd = m({'points': torch.from_numpy(make_points()).to(cuda), 'batch_size': 2})
    assert np.array_equal(bd['voxel_coords'].cpu().numpy(), g['voxel_coords'])
    feat = bd['pillar_features']
    ref = g['pillar_features']
    assert_close_per_channel(feat.detach().cpu().numpy(), ref, rtol=1e-3, atol_frac=1e-4, name="pillar_features")
    w = torch.from_numpy(np.random.default_rng(3).normal(size=ref.shape).astype(np.float32)).to(cuda)
    (feat * w).sum().backward()
    for name, grad in (("grad_linear0", m.pfn_layers[0].linear.weight.grad), ("grad_linear1", m.pfn_layers[1].linear.weight.grad)):
        assert_close_per_channel(grad.cpu().numpy(), g[name], rtol=1e-3, atol_frac=1e-4, name=name)
    np.testing.assert_allclose(m.pfn_layers[0].norm.running_mean.cpu().numpy(), g['running_mean0'], rtol=1e-3, atol=1e-5)
