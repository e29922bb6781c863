"""BASELINE configs 4 and 5 at their full size (inputs: tests/config_inputs.py), each run through the registered detector and checked against
the oracle where the oracle can follow (indices bit-exact, features element-wise per channel, tests/tolerances.py) and through size-independent
properties elsewhere.  The reference cannot run these (spconv and the CUDA ops are absent): parity for the sparse stages is
against oracle/spconv.py (unpinned restatement, cross-checked against torch conv3d in tests/test_spconv.py)."""
import numpy as np
import pytest
import torch

import config_inputs as ci
from oracle import boxes as ob, hard_voxelize as ohv, pointnet2 as op2, spconv as osp
from seeding import seeded_state_dict
from tolerances import assert_close_per_channel
from seevcn_amd.pcdet import model_cfgs as C


# ------------------------------------------------------------------------------------------ config 5: CenterPoint, VoxelResBackBone8x, 300 k points
@pytest.mark.gpu
def test_hip_centerpoint_res_backbone_at_config5_size(cuda, hip_lib):
    from seevcn_amd.pcdet.models import detectors
    from seevcn_amd.pcdet.ops import voxel_ops
    pts, gt = ci.centerpoint_scene(n_az=1200)
    assert 280_000 <= len(pts) <= 330_000, len(pts)
    pts = pts[np.random.default_rng(0).permutation(len(pts))]                      # DataProcessor.shuffle_points (data_processor.py:146-154)
    grid = np.round((np.array(ci.NUSC_RANGE[3:]) - np.array(ci.NUSC_RANGE[:3])) / np.array(ci.NUSC_VOXEL)).astype(np.int64)
    assert list(grid) == [1440, 1440, 40]
    # D1 at full size: GPU hard voxeliser == sequential first-come oracle, bit-exact (coords, slot order, counts)
    vox, crd, nmp, nv = voxel_ops.voxelize_hard(torch.from_numpy(pts).to(cuda), 0, 3, [len(pts)], ci.NUSC_RANGE, ci.NUSC_VOXEL, grid, 10, 120000)
    ov, oc, on = ohv.points_to_voxel(pts, ci.NUSC_VOXEL, ci.NUSC_RANGE, 10, 120000)
    n = int(nv[0])
    assert n == len(oc) and np.array_equal(crd[0, :n].cpu().numpy(), oc) and np.array_equal(nmp[0, :n].cpu().numpy(), on)
    assert np.array_equal(vox[0, :n].cpu().numpy(), ov) and int(on.max()) == 10            # the 10-points-per-voxel cap is exercised
    coords = torch.cat([torch.zeros((n, 1), dtype=torch.int32, device=cuda), crd[0, :n]], dim=1)

    ds = C.SyntheticDatasetInfo(class_names=C.NUSC_CLASS_NAMES, point_cloud_range=ci.NUSC_RANGE, voxel_size=ci.NUSC_VOXEL, num_point_features=3)
    net = detectors.build_detector(C.centerpoint_model_cfg(), num_class=10, dataset=ds)
    assert type(net.backbone_3d).__name__ == "VoxelResBackBone8x" and type(net.dense_head).__name__ == "CenterHead"
    sd = seeded_state_dict(net, seed=21)
    net.load_state_dict(sd)
    net = net.to(cuda)
    gt10 = np.concatenate([gt[:, :7], np.zeros((len(gt), 2), np.float32), gt[:, 7:8]], axis=1)[None]          # [box7, vx, vy, class]
    batch = {"batch_size": 1, "voxels": vox[0, :n].contiguous(), "voxel_coords": coords, "voxel_num_points": nmp[0, :n].contiguous(),
             "gt_boxes": torch.from_numpy(gt10).to(cuda)}

    # eval: every level of the residual backbone against the oracle chain (eval-mode BN), coordinates bit-exact
    net.eval()
    with torch.no_grad():
        bd = dict(batch)
        for m in net.module_list[:2]:                                              # MeanVFE, VoxelResBackBone8x
            bd = m(bd)
    feats = (ov.astype(np.float64).sum(1) / np.maximum(on, 1)[:, None]).astype(np.float32)
    np.testing.assert_allclose(bd["voxel_features"].cpu().numpy(), feats, rtol=1e-6, atol=1e-6)
    bsd = {k[len("backbone_3d."):]: v.numpy() for k, v in sd.items() if k.startswith("backbone_3d.")}
    ref = osp.voxel_res_backbone8x_forward(bsd, bd["voxel_features"].cpu().numpy(), coords.cpu().numpy(), 1, net.backbone_3d.sparse_shape)
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        t = bd["multi_scale_3d_features"][name]
        f, c, shape = ref[name]
        assert list(t.spatial_shape) == list(shape) and np.array_equal(t.indices.cpu().numpy(), c), name
        assert_close_per_channel(t.features.cpu().numpy(), f, name=name)
    t = bd["encoded_spconv_tensor"]
    f, c, shape = ref["out"]
    assert list(t.spatial_shape) == [2, 180, 180] == list(shape) and np.array_equal(t.indices.cpu().numpy(), c)
    assert_close_per_channel(t.features.cpu().numpy(), f, name="conv_out")
    assert bd["multi_scale_3d_features"]["x_conv4"].features.shape[1] == 128           # the 128 -> 128 residual layers ran

    # whole detector: train step (targets, losses, backward through the residual blocks) and eval decode
    net.train()
    ret, tb, _ = net(dict(batch))
    assert torch.isfinite(ret["loss"]) and "rpn_loss" in tb
    ret["loss"].backward()
    for key in ("conv4.1.conv1.weight", "conv4.2.conv2.weight", "conv1.0.conv1.bias", "conv_out.0.weight"):
        g = dict(net.backbone_3d.named_parameters())[key].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0, key
    net.eval()
    with torch.no_grad():
        preds, recall = net(dict(batch))
    assert len(preds) == 1 and preds[0]["pred_boxes"].shape[1] >= 7 and "gt" in recall
    # linearity of one 128 -> 128 submanifold layer at this size (size-independent property): conv(a x + b y) = a conv(x) + b conv(y)
    import seevcn_amd.spconv as spconv
    x4 = bd["multi_scale_3d_features"]["x_conv4"]
    conv = net.backbone_3d.conv4[1].conv1
    xa, xb = torch.randn_like(x4.features), torch.randn_like(x4.features)
    with torch.no_grad():
        mk = lambda ft: spconv.SparseConvTensor(ft, x4.indices, x4.spatial_shape, 1)
        ya, yb, yc = conv(mk(xa)).features, conv(mk(xb)).features, conv(mk(2.0 * xa - 3.0 * xb)).features
        bias = conv.bias.detach()
    np.testing.assert_allclose((yc - bias).cpu().numpy(), (2.0 * (ya - bias) - 3.0 * (yb - bias)).cpu().numpy(), rtol=1e-3, atol=2e-3)


# ------------------------------------------------------------------------------------------ config 4: PV-RCNN train step, DA geometry, 4096 keypoints
@pytest.mark.gpu
def test_hip_pvrcnn_train_step_at_config4_size(cuda, hip_lib):
    """4 scenes per GPU, 360-degree clouds with VCN-completed objects pasted in, sparse shape [41,1504,1504], NUM_KEYPOINTS 4096, sources
    bev / x_conv3 / x_conv4 / raw_points, 9000 -> 512 proposals, 128 RoIs x 216 grid points (source-nuscenes/pvrcnn.yaml)."""
    import seevcn_amd.synth as synth
    import seevcn_amd.vcn as V
    from seevcn_amd.pcdet.models import detectors
    from seevcn_amd.pcdet.ops.iou3d_nms import iou3d_nms_utils
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as raw
    from seevcn_amd.vcn.scene_merge import complete_scene_batch_device
    from seevcn_amd.vcn.utils import sampling
    B = 4
    pts, gt = ci.pvrcnn_scene_batch(B, n_az=350)
    per_scene = np.bincount(pts[:, 0].astype(int), minlength=B)
    assert per_scene.min() > 15000, per_scene
    # stage A feeds stage B: 16 cropped objects completed by VCN_VC (+ surface select + largest cluster), pasted into their scenes
    objs, _ = synth.make_object_batch(16, seed=1000)
    objs[:, :, 2] += 1.8                                                                 # SHIFT_COOR
    vcn = V.MODELS.build({"NAME": "VCN_VC"})
    vcn.load_state_dict(seeded_state_dict(vcn, seed=0))
    vcn = vcn.to(cuda).eval()
    with torch.no_grad():
        x = torch.from_numpy(objs).to(cuda)
        coarse = vcn({"input": x})["coarse"]
        surface, _ = sampling.get_partial_mesh_batch_device(x, coarse, k=30)
        clustered, _ = sampling.get_largest_cluster_batch_device(surface, eps=0.4, min_points=2)
        merged = complete_scene_batch_device(torch.from_numpy(pts).to(cuda), clustered, torch.arange(16, device=cuda).float() // 4, 0.1, compact=True)
    order = torch.argsort(merged[:, 0], stable=True)                                   # collate layout: stacked scene by scene
    points = merged[order].contiguous()
    assert points.shape[0] > pts.shape[0] - 16 * 1024

    ds = C.SyntheticDatasetInfo(class_names=["car"], point_cloud_range=C.DA_RANGE, voxel_size=C.DA_VOXEL, num_point_features=3)
    assert list(ds.grid_size) == [1504, 1504, 40]
    net = detectors.build_detector(C.see_pvrcnn_model_cfg(), num_class=1, dataset=ds)
    net.load_state_dict(seeded_state_dict(net, seed=6))
    net = net.to(cuda).train()
    assert net.backbone_3d.sparse_shape == [41, 1504, 1504]
    gtb = gt.copy()
    gtb[:, :, 7] = (gtb[:, :, 3] > 0)                                                   # one class
    batch = {"batch_size": B, "points": points, "gt_boxes": torch.from_numpy(gtb).to(cuda)}
    np.random.seed(0)
    torch.manual_seed(0)
    bd = dict(batch)
    for m in net.module_list:
        bd = m(bd)
    ret, tb, _ = net.get_training_loss()
    assert torch.isfinite(ret) and {"rpn_loss", "point_loss_cls", "rcnn_loss"} <= set(tb) | set(_)

    # D14 at size: 4096 keypoints per scene, index-exact against the oracle's restatement of the CUDA kernel (tie rule included)
    P = points.cpu().numpy()
    cnt = np.bincount(P[:, 0].astype(int), minlength=B)
    kp = bd["point_coords"].cpu().numpy()
    assert kp.shape == (B * 4096, 4)
    starts = np.cumsum(cnt) - cnt
    for b in (0, B - 1):                                                                # two scenes: the oracle takes ~10 s each
        want = op2.farthest_point_sampling(P[starts[b]:starts[b] + cnt[b], 1:4], 4096)
        assert np.array_equal(kp[b * 4096:(b + 1) * 4096, 1:4], P[starts[b] + want, 1:4]), b
    # D16 at size: ball query of the x_conv3 source (r 1.2 / 2.4, ns 16 / 32) vs oracle on a slice of the queries
    from seevcn_amd.pcdet.utils import common_utils
    t3 = bd["multi_scale_3d_features"]["x_conv3"]
    xyz3 = common_utils.get_voxel_centers(t3.indices[:, 1:4], downsample_times=4, voxel_size=C.DA_VOXEL, point_cloud_range=C.DA_RANGE).contiguous()
    c3 = torch.bincount(t3.indices[:, 0].long(), minlength=B).int()
    new_xyz = bd["point_coords"][:, 1:4].contiguous()
    qcnt = torch.full((B,), 4096, dtype=torch.int32, device=cuda)
    for radius, ns in ((1.2, 16), (2.4, 32)):
        idx = torch.zeros((new_xyz.shape[0], ns), dtype=torch.int32, device=cuda)
        raw.ball_query_wrapper(B, new_xyz.shape[0], radius, ns, new_xyz, qcnt, xyz3, c3, idx)
        sl = slice(4096, 4096 + 300)                                                    # 300 queries of scene 1
        c3n = c3.cpu().numpy()
        s1 = int(c3n[0])
        want = op2.ball_query(radius, ns, xyz3[s1:s1 + int(c3n[1])].cpu().numpy(), [int(c3n[1])], new_xyz[sl].cpu().numpy(), [300])
        assert np.array_equal(idx[sl].cpu().numpy(), want), radius
        # D16 features at size: the training-mode set-abstraction kernels (gather, MFMA MLP, batch-statistics BatchNorm, max) of this scale,
        # ALL 16 384 queries, against the float64 oracle on the same neighbour lists (the statistics need every row)
        k = 0 if ns == 16 else 1
        conv1, bn1, _, conv2, bn2, _ = list(net.pfe.SA_layers[0].mlps[k])
        n = lambda p: p.detach().cpu().numpy()
        row_start = torch.repeat_interleave(torch.cumsum(c3, 0) - c3, 4096).int().cpu().numpy()
        want_f = op2.sa_scale_train(xyz3.cpu().numpy(), t3.features.detach().cpu().numpy(), new_xyz.cpu().numpy(), idx.cpu().numpy(), row_start,
                                    n(conv1.weight).reshape(64, 67), n(bn1.weight), n(bn1.bias), n(conv2.weight).reshape(64, 64), n(bn2.weight), n(bn2.bias),
                                    eps=bn1.eps)
        got_f = bd["point_features_before_fusion"][:, 288 + 64 * k:288 + 64 * (k + 1)].detach().cpu().numpy()
        assert_close_per_channel(got_f, want_f, rtol=1e-3, atol_frac=1e-4, name=f"x_conv3 set-abstraction features, radius {radius}")
    # D19 at size: the 9000 -> 512 proposal NMS of scene 0 against the oracle sweep
    scores = bd["batch_cls_preds"][0].detach().max(dim=1)[0]
    boxes = bd["batch_box_preds"][0].detach()
    top = torch.topk(scores, k=9000)[1]
    keep, _ = iou3d_nms_utils.nms_gpu(boxes[top], scores[top], 0.8)
    sb, ss = boxes[top].cpu().numpy(), scores[top].cpu().numpy()
    o = np.argsort(-ss, kind="stable")
    if len(np.unique(ss)) == len(ss):
        assert np.array_equal(keep.cpu().numpy(), o[ob.nms(sb[o], 0.8)])
    assert tuple(net.roi_head.forward_ret_dict["rois"].shape) == (B, 128, 7)
    assert bd["point_features"].shape == (B * 4096, 128) and bd["point_features_before_fusion"].shape[1] == 256 + 128 + 128 + 32
    ret.backward()
    for name in ("backbone_3d.conv3.1.0.weight", "pfe.SA_layers.0.mlps.1.0.weight", "pfe.SA_rawpoints.mlps.0.0.weight", "roi_head.shared_fc_layer.0.weight",
                 "roi_head.roi_grid_pool_layer.mlps.0.0.weight", "dense_head.conv_box.weight"):
        g = dict(net.named_parameters())[name].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0, name
    net.eval()
    with torch.no_grad():
        preds, recall = net(dict(batch))
    assert len(preds) == B and all(p["pred_boxes"].shape[1] == 7 for p in preds) and "gt" in recall
