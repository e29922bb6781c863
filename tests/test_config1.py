"""BASELINE.json configs[0] as written: the reference's demo scene (demo/demo_data/pcd/000001.pcd), 4 cropped objects, VCN forward +
PointPillars -- this repo's registered modules against tests/golden/config1_demo.npz, which holds the scene (data) and the outputs of the
reference's own ResamplePoints / VCN_VC / get_partial_mesh_batch / PointPillar detector on it (tests/golden/make_config1_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import hard_voxelize as ohv
from seeding import seeded_state_dict
from seevcn_amd.pcdet import model_cfgs as C
from tolerances import assert_close_per_channel


def _fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "config1_demo.npz"))
    ends = np.cumsum(g["crop_sizes"])
    crops = [g["points"][g["crop_index"][e - n:e]] for n, e in zip(g["crop_sizes"], ends)]
    return g, crops


def test_config1_fixture_and_resample_match_the_reference(golden_dir):
    """The scene is the demo cloud (point count and checksum of tests/golden/pcd_sample.npz, made from the same file by hand), and V1
    ResamplePoints with the generator's RNG seed reproduces the reference's 1024-point inputs exactly."""
    from seevcn_amd.vcn.datasets.data_transforms import ResamplePoints
    g, crops = _fixture(golden_dir)
    pcd = np.load(os.path.join(golden_dir, "pcd_sample.npz"))
    assert len(g["points"]) == int(pcd["n_points"]) == 26715
    assert np.array_equal(g["points"][:256], pcd["head"]) and np.allclose(g["points"].astype(np.float64).sum(0), pcd["checksum"])
    np.random.seed(11)
    res = ResamplePoints({"n_points": 1024})
    mine = np.stack([res(c.astype(np.float32)) for c in crops]).astype(np.float32)
    assert np.array_equal(mine, g["vcn_input"])
    # the oracle voxeliser on the scene gives the voxel list the golden detector outputs were made from
    v, c, n = ohv.points_to_voxel(g["points"], C.PP_VOXEL["VOXEL_SIZE"], C.PP_RANGE, 32, 40000)
    assert np.array_equal(c, g["voxel_coords"][:, 1:]) and np.array_equal(n, g["voxel_num_points"])
    assert np.allclose(v.astype(np.float64).sum((0, 1)), g["voxel_checksum"])


@pytest.mark.gpu
def test_hip_config1_vcn_on_the_demo_crops(golden_dir, cuda, hip_lib):
    import seevcn_amd.vcn as V
    from seevcn_amd.vcn.utils import sampling
    g, _ = _fixture(golden_dir)
    net = V.MODELS.build({"NAME": "VCN_VC"})
    net.load_state_dict(seeded_state_dict(net, seed=0))
    net = net.to(cuda).eval()
    x = torch.from_numpy(g["vcn_input"]).to(cuda)
    with torch.no_grad():
        ret = net({"input": x})
    for k in ("coarse", "reg_rot", "reg_centre"):
        assert_close_per_channel(ret[k].cpu().numpy(), g[k], rtol=1e-3, atol_frac=1e-4, name="config1 " + k)
    # surface selection on the REFERENCE's coarse output: index sets must be identical -> bit-exact points
    surface, _ = sampling.get_partial_mesh_batch_device(x, torch.from_numpy(g["coarse"]).to(cuda), k=30)
    assert np.array_equal(surface.cpu().numpy(), g["surface"])
    # and on this build's own coarse output (differs in the last bits): the selected sets may differ only where two distances nearly tie.
    # One index more or less re-orders the whole CPython-set walk and the coordinates differ in their last bits, so the comparison is between
    # the sets of selected ROWS of each object (a surface point is an exact copy of a row of the coarse cloud it was selected from).
    mine, _ = sampling.get_partial_mesh_batch_device(x, ret["coarse"], k=30)
    my_coarse = ret["coarse"].cpu().numpy()

    def rows_of(points, cloud):
        where = {tuple(r): i for i, r in enumerate(cloud.tolist())}
        return {where[tuple(r)] for r in points.tolist()}

    for b in range(len(g["surface"])):
        want = rows_of(g["surface"][b], g["coarse"][b])
        got = rows_of(mine[b].cpu().numpy(), my_coarse[b])
        assert len(want & got) >= 0.98 * len(want | got), (b, len(want), len(got), len(want & got))


@pytest.mark.gpu
def test_hip_config1_pointpillar_detector_on_the_demo_scene(golden_dir, cuda, hip_lib):
    """The registered PointPillar detector (detectors/pointpillar.py:4-37) end to end in eval mode: HIP hard voxeliser -> PillarVFE ->
    PointPillarScatter -> BaseBEVBackbone -> AnchorHeadSingle -> post_processing."""
    from seevcn_amd.pcdet.models import detectors
    from seevcn_amd.pcdet.ops import voxel_ops
    g, _ = _fixture(golden_dir)
    vs, rng_ = C.PP_VOXEL["VOXEL_SIZE"], C.PP_RANGE
    grid = np.round((np.array(rng_[3:]) - np.array(rng_[:3])) / np.array(vs)).astype(np.int64)
    assert list(grid) == [432, 496, 1]
    pts = torch.from_numpy(g["points"]).to(cuda)
    vox, crd, nmp, nv = voxel_ops.voxelize_hard(pts, 0, 3, [len(pts)], rng_, vs, grid, 32, 40000)
    n = int(nv[0])
    assert n == len(g["voxel_coords"]) and np.array_equal(crd[0, :n].cpu().numpy(), g["voxel_coords"][:, 1:])
    assert np.array_equal(nmp[0, :n].cpu().numpy(), g["voxel_num_points"])
    assert np.allclose(vox[0, :n].double().sum((0, 1)).cpu().numpy(), g["voxel_checksum"])
    coords = torch.cat([torch.zeros((n, 1), dtype=torch.int32, device=cuda), crd[0, :n]], dim=1)

    ds = C.SyntheticDatasetInfo(point_cloud_range=rng_, voxel_size=vs, num_point_features=3)
    net = detectors.build_detector(C.pointpillar_model_cfg(), num_class=3, dataset=ds)
    assert type(net).__name__ == "PointPillar" and [type(m).__name__ for m in net.module_list] == ["PillarVFE", "PointPillarScatter", "BaseBEVBackbone",
                                                                                                  "AnchorHeadSingle"]
    net.load_state_dict(seeded_state_dict(net, seed=31))
    net = net.to(cuda).eval()
    seen = {}
    for m in net.module_list:
        m.register_forward_hook(lambda mod, i, o, seen=seen: seen.update({type(mod).__name__: dict(o)}))
    with torch.no_grad():
        preds, recall = net({"batch_size": 1, "voxels": vox[0, :n].contiguous(), "voxel_coords": coords, "voxel_num_points": nmp[0, :n].contiguous()})
    assert_close_per_channel(seen["PillarVFE"]["pillar_features"][::4].cpu().numpy(), g["pillar_features"], rtol=1e-3, atol_frac=1e-4, name="pillar_features")
    sf = seen["BaseBEVBackbone"]["spatial_features_2d"]
    assert list(sf.shape) == list(g["sf2d_shape"]) == [1, 384, 248, 216]
    assert_close_per_channel(sf[0, :, ::8, ::8].cpu().numpy(), g["sf2d_sample"], rtol=1e-3, atol_frac=1e-4, name="spatial_features_2d", channel_axis=0)
    np.testing.assert_allclose(sf.double().sum((0, 2, 3)).cpu().numpy(), g["sf2d_channel_sum"], rtol=1e-3, atol=1e-2 * np.abs(g["sf2d_channel_sum"]).max())
    hd = seen["AnchorHeadSingle"]
    assert hd["batch_box_preds"].shape[1] == int(g["n_anchors"]) == 248 * 216 * 6
    pick = torch.from_numpy(g["anchor_pick"]).to(cuda)
    assert_close_per_channel(hd["batch_cls_preds"][0, pick].cpu().numpy(), g["cls_preds"], rtol=1e-3, atol_frac=1e-4, name="batch_cls_preds")
    assert_close_per_channel(hd["batch_box_preds"][0, pick].cpu().numpy(), g["box_preds"], rtol=1e-3, atol_frac=1e-4, name="batch_box_preds")
    # final boxes: NMS at 0.01 over scores that differ in the last bits between the CPU and GPU convolutions -> compare as sets
    pb, ps = preds[0]["pred_boxes"].cpu().numpy(), preds[0]["pred_scores"].cpu().numpy()
    gb, gs = g["pred_boxes"], g["pred_scores"]
    assert abs(len(pb) - len(gb)) <= 5
    hit = sum((np.abs(pb - gb[i]).max(1) + np.abs(ps - gs[i])).min() < 5e-3 for i in range(len(gb)))
    assert hit >= 0.97 * len(gb), (hit, len(gb))
