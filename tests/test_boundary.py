"""The pybind-level extension modules and spconv.utils generators the reference's own Python binds: each test calls
through the same call sequence the reference's wrapper uses (cited) and checks against the oracle."""
import numpy as np
import pytest
import torch

from oracle import boxes as ob, chamfer as och, hard_voxelize as ohv, pool_ops as opo


def _rand_boxes(rng, n, spread):
    return np.concatenate([rng.uniform(-spread, spread, (n, 2)), rng.uniform(-1, 1, (n, 1)), rng.uniform(1.0, 5.0, (n, 3)),
                           rng.uniform(-3.1, 3.1, (n, 1))], 1).astype(np.float32)


def test_oracle_points_in_boxes_cpu_hand_case():
    rois = np.array([[0, 0, 0, 4, 2, 2, 0.0]], np.float32)
    pts = np.array([[-1.9, -0.9, -0.9], [1.9, 0.9, 0.9], [2.005, 0.0, 0.0], [5, 0, 0]], np.float32)     # third point: inside only through the 1e-2 margin
    assert opo.points_in_boxes_cpu(pts, rois).tolist() == [[1, 1, 1, 0]]


def test_boundary_modules_export_the_reference_names():
    from seevcn_amd.pcdet.ops.iou3d_nms import iou3d_nms_cuda
    from seevcn_amd.pcdet.ops.roiaware_pool3d import roiaware_pool3d_cuda
    from seevcn_amd.vcn.extensions.chamfer_dist import chamfer
    import seevcn_amd.spconv as spconv
    for name in ("boxes_overlap_bev_gpu", "boxes_iou_bev_gpu", "nms_gpu", "nms_normal_gpu", "boxes_iou_bev_cpu"):     # iou3d_nms_api.cpp:12-17
        assert callable(getattr(iou3d_nms_cuda, name))
    for name in ("forward", "backward", "points_in_boxes_gpu", "points_in_boxes_cpu"):                              # roiaware_pool3d.cpp:172-177
        assert callable(getattr(roiaware_pool3d_cuda, name))
    assert callable(chamfer.forward) and callable(chamfer.backward)                                                 # chamfer_cuda.cpp:36-39
    for name in ("VoxelGeneratorV2", "VoxelGenerator", "Point2VoxelCPU3d"):                                         # data_processor.py:17-26
        assert isinstance(getattr(spconv.utils, name), type)
    with pytest.raises(NotImplementedError):
        roiaware_pool3d_cuda.forward()


@pytest.mark.gpu
@pytest.mark.parametrize("n,thr", [(64, 0.1), (1500, 0.7)])
def test_hip_iou3d_nms_cuda_module_through_the_reference_call_sequence(cuda, hip_lib, n, thr):
    """the body of the reference's nms_gpu / boxes_iou_bev (iou3d_nms_utils.py:33-45,84-99) over OUR iou3d_nms_cuda"""
    from seevcn_amd.pcdet.ops.iou3d_nms import iou3d_nms_cuda, iou3d_nms_utils as u
    rng = np.random.default_rng(n)
    b = _rand_boxes(rng, n, 25.0)
    b[: n // 3, 0:2] = b[n // 3: 2 * (n // 3), 0:2][: n // 3] + rng.normal(0, 0.3, (n // 3, 2))
    s = rng.uniform(size=n).astype(np.float32)
    boxes, scores = torch.from_numpy(b).to(cuda), torch.from_numpy(s).to(cuda)
    for fn, normal in ((iou3d_nms_cuda.nms_gpu, False), (iou3d_nms_cuda.nms_normal_gpu, True)):
        order = scores.sort(0, descending=True)[1]
        sb = boxes[order].contiguous()
        keep = torch.LongTensor(sb.size(0))
        num_out = fn(sb, keep, thr)
        assert isinstance(num_out, int)
        got = order[keep[:num_out].to(cuda)].contiguous().cpu().numpy()
        o = np.argsort(-s, kind="stable")
        assert np.array_equal(got, o[ob.nms(b[o], thr, normal=normal)])
    assert np.array_equal(got, u.nms_normal_gpu(boxes, scores, thr)[0].cpu().numpy())
    a2 = _rand_boxes(rng, 33, 25.0)
    ta = torch.from_numpy(a2).to(cuda)
    ans = torch.zeros((33, n), dtype=torch.float32, device=cuda)
    assert iou3d_nms_cuda.boxes_iou_bev_gpu(ta.contiguous(), boxes.contiguous(), ans) == 1
    np.testing.assert_allclose(ans.cpu().numpy(), ob.boxes_iou_bev(a2, b), rtol=1e-3, atol=1e-4)
    ovl = torch.zeros((33, n), dtype=torch.float32, device=cuda)
    assert iou3d_nms_cuda.boxes_overlap_bev_gpu(ta.contiguous(), boxes.contiguous(), ovl) == 1
    np.testing.assert_allclose(ovl.cpu().numpy(), ob.boxes_overlap_bev(a2, b), rtol=1e-3, atol=2e-4)
    cpu_ans = torch.zeros((33, n))
    assert iou3d_nms_cuda.boxes_iou_bev_cpu(torch.from_numpy(a2), torch.from_numpy(b), cpu_ans) == 1     # iou3d_nms_utils.py:11-28
    assert torch.equal(cpu_ans, ans.cpu())
    with pytest.raises(RuntimeError):
        iou3d_nms_cuda.boxes_iou_bev_gpu(ta.cpu(), boxes, ans)                                           # CHECK_CUDA (iou3d_nms.cpp:14-26)


@pytest.mark.gpu
def test_hip_roiaware_pool3d_cuda_points_in_boxes(cuda, hip_lib):
    """roiaware_pool3d_utils.py:9-41 call sequences over OUR roiaware_pool3d_cuda"""
    from seevcn_amd.pcdet.ops.roiaware_pool3d import roiaware_pool3d_cuda, roiaware_pool3d_utils as r
    rng = np.random.default_rng(3)
    pts = rng.uniform(-12, 12, (2, 3000, 3)).astype(np.float32)
    pts[:, :, 2] *= 0.15
    boxes = np.stack([_rand_boxes(rng, 30, 10.0) for _ in range(2)])
    box_idxs = torch.from_numpy(pts).to(cuda).new_zeros((2, 3000), dtype=torch.int).fill_(-1)
    assert roiaware_pool3d_cuda.points_in_boxes_gpu(torch.from_numpy(boxes).to(cuda).contiguous(), torch.from_numpy(pts).to(cuda).contiguous(), box_idxs) == 1
    assert np.array_equal(box_idxs.cpu().numpy(), ob.points_in_boxes(pts, boxes))
    assert np.array_equal(r.points_in_boxes_gpu(torch.from_numpy(pts).to(cuda), torch.from_numpy(boxes).to(cuda)).cpu().numpy(), box_idxs.cpu().numpy())
    p1, b1 = torch.from_numpy(pts[0, :400]), torch.from_numpy(boxes[0, :9])
    point_indices = p1.new_zeros((9, 400), dtype=torch.int)
    assert roiaware_pool3d_cuda.points_in_boxes_cpu(b1.float().contiguous(), p1.float().contiguous(), point_indices) == 1
    want = opo.points_in_boxes_cpu(pts[0, :400], boxes[0, :9])
    assert np.array_equal(point_indices.numpy(), want) and want.sum() > 5
    m = r.points_in_boxes_cpu(pts[0, :400], boxes[0, :9])
    assert isinstance(m, np.ndarray) and np.array_equal(m, want)


@pytest.mark.gpu
def test_hip_chamfer_module_forward_backward_lists(cuda, hip_lib):
    """chamfer_dist/__init__.py:13-25 over OUR chamfer module"""
    from seevcn_amd.vcn.extensions.chamfer_dist import chamfer, ChamferDistanceL2
    rng = np.random.default_rng(5)
    x1, x2 = rng.normal(size=(3, 257, 3)).astype(np.float32), rng.normal(size=(3, 190, 3)).astype(np.float32)
    t1, t2 = torch.from_numpy(x1).to(cuda), torch.from_numpy(x2).to(cuda)
    out = chamfer.forward(t1, t2)
    assert isinstance(out, list) and len(out) == 4 and out[2].dtype == torch.int32
    d1, d2, i1, i2 = och.forward(x1, x2)
    assert np.array_equal(out[2].cpu().numpy(), i1) and np.array_equal(out[3].cpu().numpy(), i2)
    np.testing.assert_allclose(out[0].cpu().numpy(), d1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out[1].cpu().numpy(), d2, rtol=1e-5, atol=1e-6)
    g1, g2 = rng.normal(size=d1.shape).astype(np.float32), rng.normal(size=d2.shape).astype(np.float32)
    gx = chamfer.backward(t1, t2, out[2], out[3], torch.from_numpy(g1).to(cuda), torch.from_numpy(g2).to(cuda))
    assert isinstance(gx, list) and len(gx) == 2
    w1, w2 = och.backward(x1, x2, i1, i2, g1, g2)
    np.testing.assert_allclose(gx[0].cpu().numpy(), w1, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gx[1].cpu().numpy(), w2, rtol=1e-4, atol=1e-5)
    a = t1.clone().requires_grad_(True)
    ChamferDistanceL2()(a, t2).backward()
    assert a.grad is not None and torch.isfinite(a.grad).all()


@pytest.mark.gpu
def test_hip_spconv_utils_voxel_generators(cuda, hip_lib):
    """VoxelGeneratorWrapper's three bindings (data_processor.py:15-60) over OUR spconv.utils"""
    import seevcn_amd.synth as synth
    import seevcn_amd.spconv as spconv
    p, _ = synth.make_scene(2003, n_az=150)
    p = np.concatenate([p, np.random.default_rng(0).uniform(size=(len(p), 1)).astype(np.float32)], 1)
    p = p[np.random.default_rng(1).permutation(len(p))]
    vs, rg, mp, mv = [0.16, 0.16, 4], [0, -39.68, -3, 69.12, 39.68, 1], 32, 900
    ov, oc, on = ohv.points_to_voxel(p, vs, rg, mp, mv)
    g2 = spconv.utils.VoxelGeneratorV2(voxel_size=vs, point_cloud_range=rg, max_num_points=mp, max_voxels=mv)
    out = g2.generate(p)
    assert isinstance(out, dict)
    assert np.array_equal(out['voxels'], ov) and np.array_equal(out['coordinates'], oc) and np.array_equal(out['num_points_per_voxel'], on)
    assert out['voxels'].dtype == np.float32 and out['coordinates'].dtype == np.int32 and len(oc) == mv       # the voxel cap is exercised
    g1 = spconv.utils.VoxelGenerator(voxel_size=vs, point_cloud_range=rg, max_num_points=mp, max_voxels=mv)
    v, c, n = g1.generate(p)
    assert np.array_equal(v, ov) and np.array_equal(c, oc) and np.array_equal(n, on)
    g3 = spconv.utils.Point2VoxelCPU3d(vsize_xyz=vs, coors_range_xyz=rg, num_point_features=4, max_num_points_per_voxel=mp, max_num_voxels=mv)
    tv_v, tv_c, tv_n = g3.point_to_voxel(p)
    assert np.array_equal(tv_v.numpy(), ov) and np.array_equal(tv_c.numpy(), oc) and np.array_equal(tv_n.numpy(), on)
    assert list(g2.grid_size) == [432, 496, 1]
    ev, ec, en = g1.generate(np.zeros((0, 4), np.float32))
    assert ev.shape == (0, mp, 4) and ec.shape == (0, 3) and en.shape == (0,)
