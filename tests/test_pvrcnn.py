"""PV-RCNN second stage: VoxelSetAbstraction, PointHeadSimple, PVRCNNHead (+ RoIHeadTemplate, ProposalTargetLayer) against goldens
produced by the reference's own classes (tests/golden/make_pvrcnn_golden.py; CUDA ops of the reference served by the oracle)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from pvrcnn_inputs import SMALL, make_inputs
from seeding import seeded_state_dict
from seevcn_amd.pcdet import model_cfgs as C

RTOL = 1e-3


def _ok(a, b, rtol=RTOL, atol_frac=1e-4, name=""):
    """element-wise |a-b| <= rtol*|b| + atol_frac * max|b[:, c]| per channel (tests/tolerances.py)"""
    from tolerances import assert_close_per_channel
    assert_close_per_channel(a, b, rtol=rtol, atol_frac=atol_frac, name=name)
    return True


def _build(cuda=None):
    from seevcn_amd.pcdet.models import dense_heads, roi_heads
    from seevcn_amd.pcdet.models.backbones_3d import pfe
    pfe_cfg, ph_cfg, rh_cfg = C.pvrcnn_cfg(**SMALL)
    vsa = pfe.__all__["VoxelSetAbstraction"](pfe_cfg, voxel_size=[0.05, 0.05, 0.1], point_cloud_range=C.KITTI_RANGE, num_bev_features=32,
                                             num_rawpoint_features=4)
    ph = dense_heads.__all__["PointHeadSimple"](num_class=1, input_channels=vsa.num_point_features_before_fusion, model_cfg=ph_cfg)
    rh = roi_heads.__all__["PVRCNNHead"](input_channels=vsa.num_point_features, model_cfg=rh_cfg, num_class=1)
    for m, seed in ((vsa, 11), (ph, 12), (rh, 13)):
        m.load_state_dict(seeded_state_dict(m, seed=seed))
        if cuda is not None:
            m.to(cuda)
    return vsa, ph, rh


def test_module_shapes_and_state_dict_keys():
    vsa, ph, rh = _build()
    assert vsa.num_point_features_before_fusion == 32 + 128 + 128 + 32 and vsa.num_point_features == 128
    keys = set(vsa.state_dict()) | set(rh.state_dict())
    assert "SA_rawpoints.mlps.0.0.weight" in keys and "SA_layers.1.mlps.1.3.weight" in keys and "vsa_point_feature_fusion.0.weight" in keys
    assert "roi_grid_pool_layer.mlps.0.0.weight" in keys and "shared_fc_layer.0.weight" in keys and "reg_layers.7.bias" in keys
    assert rh.shared_fc_layer[0].weight.shape == (256, 6 * 6 * 6 * 128, 1)           # 27648 -> 256


@pytest.mark.gpu
def test_hip_pvrcnn_heads_match_reference_golden(golden_dir, cuda, hip_lib):
    g = np.load(os.path.join(golden_dir, "pvrcnn_heads.npz"))
    inp = make_inputs()
    vsa, ph, rh = _build(cuda)
    t = lambda a: torch.from_numpy(a).to(cuda)

    def batch():
        return {"batch_size": 2, "points": t(inp["points"]), "gt_boxes": t(inp["gt_boxes"]), "spatial_features": t(inp["spatial_features"]),
                "spatial_features_stride": 8,
                "multi_scale_3d_features": {k: SimpleNamespace(indices=t(inp[k + "_indices"]), features=t(inp[k + "_features"])) for k in ("x_conv3", "x_conv4")},
                "batch_cls_preds": t(inp["batch_cls_preds"]), "batch_box_preds": t(inp["batch_box_preds"]), "cls_preds_normalized": False}

    for m in (vsa, ph, rh):
        m.train()
    np.random.seed(7)
    torch.manual_seed(7)
    bd = rh(ph(vsa(batch())))
    assert np.array_equal(bd["point_coords"].cpu().numpy(), g["point_coords"])                      # FPS keypoints: index-exact
    assert _ok(bd["point_features_before_fusion"].detach().cpu().numpy(), g["point_features_before_fusion"], name="point_features_before_fusion")
    assert _ok(bd["point_features"].detach().cpu().numpy(), g["point_features"], atol_frac=5e-4, name="point_features (train-mode BN over 512 rows)")
    assert np.array_equal(ph.forward_ret_dict["point_cls_labels"].cpu().numpy(), g["point_cls_labels"])
    point_loss, tb1 = ph.get_loss()
    assert abs(point_loss.item() - float(g["point_loss"])) < RTOL * abs(float(g["point_loss"])) and tb1["point_pos_num"] == float(g["point_pos_num"])
    fr = rh.forward_ret_dict
    np.testing.assert_allclose(fr["rois"].cpu().numpy(), g["train_rois"], rtol=0, atol=0)           # same NMS survivors, same random sample
    np.testing.assert_allclose(fr["gt_iou_of_rois"].cpu().numpy(), g["gt_iou_of_rois"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(fr["gt_of_rois"].cpu().numpy(), g["gt_of_rois"], rtol=1e-4, atol=1e-4)
    assert np.array_equal(fr["reg_valid_mask"].cpu().numpy(), g["reg_valid_mask"])
    np.testing.assert_allclose(fr["rcnn_cls_labels"].cpu().numpy(), g["rcnn_cls_labels"], rtol=1e-3, atol=1e-4)
    # rcnn_* sit behind train-mode BatchNorm over only 64 RoIs and a 27648-term fp32 contraction: the batch statistics amplify last-bit
    # differences of the GPU GEMM order; 1e-3 relative + 1e-3 of each output's largest value
    assert _ok(fr["rcnn_cls"].detach().cpu().numpy(), g["rcnn_cls"], atol_frac=1e-3, name="rcnn_cls")
    assert _ok(fr["rcnn_reg"].detach().cpu().numpy(), g["rcnn_reg"], atol_frac=1e-3, name="rcnn_reg")
    loss, tb2 = rh.get_loss()
    for k in ("rcnn_loss_cls", "rcnn_loss_reg", "rcnn_loss_corner", "rcnn_loss"):
        assert abs(tb2[k] - float(g[k])) < 1e-3 * abs(float(g[k])) + 1e-4, (k, tb2[k], float(g[k]))
    (point_loss + loss).backward()
    assert torch.isfinite(vsa.vsa_point_feature_fusion[0].weight.grad).all() and torch.isfinite(rh.shared_fc_layer[0].weight.grad).all()
    # eval
    for m in (vsa, ph, rh):
        m.eval()
    with torch.no_grad():
        bd = rh(ph(vsa(batch())))
    np.testing.assert_allclose(bd["rois"].cpu().numpy(), g["eval_rois"], rtol=0, atol=0)
    assert np.array_equal(bd["roi_labels"].cpu().numpy(), g["eval_roi_labels"])
    assert _ok(bd["point_features"].cpu().numpy(), g["eval_point_features"], name="eval point_features")
    assert _ok(bd["batch_cls_preds"].cpu().numpy(), g["eval_batch_cls_preds"], atol_frac=1e-3, name="eval batch_cls_preds")
    assert _ok(bd["batch_box_preds"].cpu().numpy(), g["eval_batch_box_preds"], name="eval batch_box_preds")


@pytest.mark.gpu
def test_hip_pvrcnn_detector_train_step(cuda, hip_lib):
    """Full PVRCNN built from the registries (reduced keypoints / RoIs): one train step and one eval pass run end to end."""
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet.models import detectors
    pts, gt = synth.make_scene_batch(2, seed=2000, n_az=100)
    cfg = C.pvrcnn_model_cfg(num_keypoints=512, roi_per_image=32, nms_post_train=128, nms_pre_train=2048)
    net = detectors.build_detector(cfg, num_class=3, dataset=C.SyntheticDatasetInfo())
    net.load_state_dict(seeded_state_dict(net, seed=6))
    net = net.to(cuda).train()
    np.random.seed(0)
    torch.manual_seed(0)
    batch = {"batch_size": 2, "points": torch.from_numpy(pts).to(cuda), "gt_boxes": torch.from_numpy(gt).to(cuda)}
    ret, tb, _ = net(dict(batch))
    assert torch.isfinite(ret["loss"]) and {"rpn_loss", "point_loss_cls", "rcnn_loss"} <= set(tb)
    ret["loss"].backward()
    assert net.backbone_3d.conv3[1][0].weight.grad is not None and torch.isfinite(net.backbone_3d.conv3[1][0].weight.grad).all()
    net.eval()
    with torch.no_grad():
        preds, recall = net(dict(batch))
    assert len(preds) == 2 and preds[0]["pred_boxes"].shape[1] == 7 and "gt" in recall
