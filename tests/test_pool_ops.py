"""RoI-aware pooling (PartA2) and the PV-RCNN++ stack ops (voxel query, vector pool, local three-nn): HIP vs the line-by-line CPU
restatement in oracle/pool_ops.py.  The reference kernels are CUDA-only and untested upstream: parity unpinned at op level."""
import numpy as np
import pytest
import torch

from oracle import pool_ops as O


def _boxes_points(rng, n_box, n_pts):
    boxes = np.concatenate([rng.uniform(-6, 6, (n_box, 2)), rng.uniform(-1, 1, (n_box, 1)), rng.uniform(1.5, 5, (n_box, 3)),
                            rng.uniform(-3.1, 3.1, (n_box, 1))], 1).astype(np.float32)
    pts = np.concatenate([rng.uniform(-8, 8, (n_pts, 2)), rng.uniform(-2, 2, (n_pts, 1))], 1).astype(np.float32)
    pts[: n_pts // 3] = (boxes[rng.integers(0, n_box, n_pts // 3), :3] + rng.normal(0, 0.7, (n_pts // 3, 3))).astype(np.float32)
    return boxes, pts


def test_oracle_roiaware_pool_small_hand_case():
    rois = np.array([[0, 0, 0, 4, 2, 2, 0.0]], np.float32)
    pts = np.array([[-1.9, -0.9, -0.9], [1.9, 0.9, 0.9], [1.8, 0.8, 0.8], [5, 0, 0]], np.float32)
    feat = np.array([[1.0], [2.0], [7.0], [100.0]], np.float32)
    pooled, argmax, idx = O.roiaware_pool3d_forward(rois, pts, feat, (2, 2, 2), 4, 0)
    assert pooled[0, 0, 0, 0, 0] == 1 and pooled[0, 1, 1, 1, 0] == 7 and argmax[0, 1, 1, 1, 0] == 2 and argmax[0, 0, 1, 0, 0] == -1
    assert list(idx[0, 1, 1, 1]) == [2, 1, 2, 0]
    avg = O.roiaware_pool3d_forward(rois, pts, feat, (2, 2, 2), 4, 1)[0]
    assert avg[0, 1, 1, 1, 0] == 4.5
    assert O.points_in_boxes_cpu(pts, rois).tolist() == [[1, 1, 1, 0]]


@pytest.mark.gpu
@pytest.mark.parametrize("out_size,max_pts,C", [((3, 4, 2), 6, 5), (2, 128, 16), ((6, 6, 6), 3, 3)])
def test_hip_roiaware_pool_forward_backward_vs_oracle(cuda, hip_lib, out_size, max_pts, C):
    from seevcn_amd.pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils as R
    rng = np.random.default_rng(C)
    boxes, pts = _boxes_points(rng, 7, 2600)
    feat = rng.normal(size=(len(pts), C)).astype(np.float32)
    osz = (out_size,) * 3 if isinstance(out_size, int) else out_size
    pool = R.RoIAwarePool3d(out_size, max_pts_each_voxel=max_pts)
    for method, name in ((0, 'max'), (1, 'avg')):
        f = torch.from_numpy(feat).to(cuda).requires_grad_(True)
        got = pool(torch.from_numpy(boxes).to(cuda), torch.from_numpy(pts).to(cuda), f, pool_method=name)
        want, argmax, idx = O.roiaware_pool3d_forward(boxes, pts, feat, osz, max_pts, method)
        assert got.shape == want.shape
        if method == 0:
            assert np.array_equal(got.detach().cpu().numpy(), want)
        else:
            np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=1e-6, atol=1e-6)
        go = rng.normal(size=want.shape).astype(np.float32)
        got.backward(torch.from_numpy(go).to(cuda))
        np.testing.assert_allclose(f.grad.cpu().numpy(), O.roiaware_pool3d_backward(idx, argmax, go, len(pts), method), rtol=1e-4, atol=1e-5)
    assert (idx[..., 0] == max_pts - 1).any() or max_pts > 100          # the per-voxel cap is exercised
    m = R.points_in_boxes_cpu(pts, boxes)
    assert isinstance(m, np.ndarray) and np.array_equal(m, O.points_in_boxes_cpu(pts, boxes)) and m.sum() > 50


def _stack_scene(rng, B=2):
    xyz_cnt = np.array([900, 700][:B], np.int32)
    new_cnt = np.array([40, 25][:B], np.int32)
    xyz = rng.uniform(-4, 4, (int(xyz_cnt.sum()), 3)).astype(np.float32)
    new_xyz = np.concatenate([xyz[s:s + n][rng.integers(0, n, m)] + rng.normal(0, 0.05, (m, 3)).astype(np.float32)
                              for s, n, m in zip(np.cumsum(xyz_cnt) - xyz_cnt, xyz_cnt, new_cnt)]).astype(np.float32)
    return xyz, xyz_cnt, new_xyz, new_cnt


@pytest.mark.gpu
@pytest.mark.parametrize("neighbor_type,pooling_type,nsample,use_xyz", [(0, 0, -1, 1), (1, 0, 12, 1), (0, 1, -1, 0), (1, 1, 5, 1)])
def test_hip_vector_pool_vs_oracle(cuda, hip_lib, neighbor_type, pooling_type, nsample, use_xyz):
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as P
    rng = np.random.default_rng(10 + neighbor_type * 2 + pooling_type)
    xyz, xyz_cnt, new_xyz, new_cnt = _stack_scene(rng)
    c_in, c_each, grid, dist = 8, 4, (2, 3, 2), 1.2
    feat = rng.normal(size=(len(xyz), c_in)).astype(np.float32)
    dev = lambda a: torch.from_numpy(a).to(cuda)
    f = dev(feat).requires_grad_(True)
    nf, nxyz, mean_pts, cnt = P.vector_pool_with_voxel_query_op(dev(xyz), dev(xyz_cnt), f, dev(new_xyz), dev(new_cnt), *grid, dist, c_each, use_xyz,
                                                                2, nsample, neighbor_type, pooling_type)   # 2 rows/query: forces the retry loop
    raw, rxyz, ocnt, rows = O.vector_pool(xyz, xyz_cnt, feat, new_xyz, new_cnt, grid, dist, c_each, use_xyz, nsample, neighbor_type, pooling_type)
    assert np.array_equal(cnt.cpu().numpy(), ocnt) and len(rows) > 100
    norm = np.maximum(ocnt[:, :, None].astype(np.float32), 1e-6)
    want = (raw.reshape(len(new_xyz), -1, c_each) / norm).reshape(len(new_xyz), -1)
    np.testing.assert_allclose(nf.detach().cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    if use_xyz:
        np.testing.assert_allclose(nxyz.cpu().numpy(), (rxyz.reshape(len(new_xyz), -1, 3) / norm).reshape(len(new_xyz), -1), rtol=1e-6, atol=1e-6)
    go = rng.normal(size=want.shape).astype(np.float32)
    nf.backward(dev(go))
    np.testing.assert_allclose(f.grad.cpu().numpy(), O.vector_pool_grad(go, ocnt, rows, len(xyz), c_in, c_each), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_hip_local_three_nn_and_voxel_query_vs_oracle(cuda, hip_lib):
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as P, voxel_query_utils as V
    rng = np.random.default_rng(21)
    xyz, xyz_cnt, new_xyz, new_cnt = _stack_scene(rng)
    dev = lambda a: torch.from_numpy(a).to(cuda)
    G = 4
    centers = (new_xyz[:, None, :] + rng.uniform(-0.5, 0.5, (len(new_xyz), G, 3))).astype(np.float32)
    for neighbor_type, nsample in ((0, -1), (1, 7)):
        dist, idx, avg = P.three_nn_for_vector_pool_by_two_step(dev(xyz), dev(xyz_cnt), dev(new_xyz), dev(centers), dev(new_cnt), 0.9, nsample,
                                                                neighbor_type, 1, G, 2.0)
        lists = O.query_stacked_local_neighbor_idxs(xyz, xyz_cnt, new_xyz, new_cnt, 0.9 * 2.0, nsample, neighbor_type)
        d2, oidx = O.query_three_nn_by_stacked_local_idxs(xyz, centers, lists)
        has = oidx[:, :, 0] >= 0
        assert has.sum() > 50 and np.array_equal(idx.cpu().numpy()[has], oidx[has])
        np.testing.assert_array_equal(dist.cpu().numpy()[has], np.sqrt(d2[has]))
        assert int(avg) == -(-sum(len(l) for l in lists) // len(new_xyz))
    # voxel query: a voxel grid over the support points (last point of a voxel wins, like a scatter)
    vs, lo = 0.5, -4.0
    coords = np.floor((xyz - lo) / vs).astype(np.int64).clip(0, 15)
    bidx = np.repeat(np.arange(len(xyz_cnt)), xyz_cnt)
    point_indices = np.full((len(xyz_cnt), 16, 16, 16), -1, np.int32)
    point_indices[bidx, coords[:, 2], coords[:, 1], coords[:, 0]] = np.arange(len(xyz), dtype=np.int32)
    nc = np.floor((new_xyz - lo) / vs).astype(np.int32).clip(0, 15)
    new_coords = np.stack([np.repeat(np.arange(len(new_cnt)), new_cnt).astype(np.int32), nc[:, 2], nc[:, 1], nc[:, 0]], 1).astype(np.int32)
    new_coords[3, 1:] = 15
    far = new_xyz.copy()
    far[3] += 100                                                              # an empty ball
    idx, empty = V.voxel_query((1, 2, 2), 0.8, 6, dev(xyz), dev(far), dev(new_coords), dev(point_indices))
    want = O.voxel_query((1, 2, 2), 0.8, 6, xyz, far, new_coords, point_indices)
    wempty = want[:, 0] == -1
    want[wempty] = 0
    assert np.array_equal(idx.cpu().numpy(), want) and np.array_equal(empty.cpu().numpy(), wempty) and wempty[3] and wempty.sum() < 10
