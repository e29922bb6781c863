"""SECOND dense path: BaseBEVBackbone + AnchorHeadSingle (anchors, fused target assignment, losses, fused decode) against the
goldens produced by the reference's own modules (tests/golden/make_head_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import heads as oh
from seeding import seeded_state_dict
from tolerances import assert_close_per_channel
from seevcn_amd.pcdet import model_cfgs as C

GRID = np.array([1408, 1600, 40])


def _golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "second_head.npz"))
    feat = (np.random.default_rng(int(g["feat_seed"])).normal(size=(2, 16, 200, 176)) * 0.5).astype(np.float32)
    return g, feat


def test_oracle_anchors_targets_decode_match_reference_golden(golden_dir):
    g, _ = _golden(golden_dir)
    cfgs = C.SECOND_DENSE_HEAD["ANCHOR_GENERATOR_CONFIG"]
    anchors, per_set = oh.generate_anchors(cfgs, GRID, C.KITTI_RANGE)
    assert anchors.shape == (211200, 7) and per_set == [2, 2, 2]
    sel = g["sel"]
    np.testing.assert_allclose(anchors[sel], g["anchors_head"], rtol=0, atol=2e-5)
    labels, targets, weights = oh.assign_targets(anchors, per_set, [1, 2, 3], [c["matched_threshold"] for c in cfgs],
                                                 [c["unmatched_threshold"] for c in cfgs], g["gt_boxes"])
    assert np.array_equal(labels[:, sel], g["box_cls_labels"])
    assert int((labels > 0).sum()) == int(g["num_pos"]) and int((labels < 0).sum()) == int(g["num_ignore"])
    np.testing.assert_allclose(targets[:, sel], g["box_reg_targets"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(weights[:, sel], g["reg_weights"])


def test_modules_state_dict_and_cpu_refusal():
    from seevcn_amd.pcdet.models import backbones_2d, dense_heads
    bb = backbones_2d.__all__["BaseBEVBackbone"](C.SECOND_BACKBONE_2D, 256)
    assert bb.num_bev_features == 512 and "blocks.0.1.weight" in bb.state_dict() and "deblocks.1.0.weight" in bb.state_dict()
    assert sum(p.numel() for p in bb.parameters()) == 4_576_768                       # BEV ~4.57 M
    head = dense_heads.__all__["AnchorHeadSingle"](model_cfg=C.SECOND_DENSE_HEAD, input_channels=512, num_class=3, class_names=C.CLASS_NAMES,
                                                   grid_size=GRID, point_cloud_range=np.array(C.KITTI_RANGE, np.float32))
    assert head.num_anchors_per_location == 6 and head.conv_box.out_channels == 42 and head.conv_dir_cls.out_channels == 12
    assert [tuple(a.shape) for a in head.anchors] == [(1, 200, 176, 1, 2, 7)] * 3


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_bev_backbone_matches_reference_golden(golden_dir, cuda, hip_lib):
    from seevcn_amd.pcdet.models import backbones_2d
    g = np.load(os.path.join(golden_dir, "bev_backbone.npz"))
    cfg = dict(C.SECOND_BACKBONE_2D, LAYER_NUMS=[2, 2], NUM_FILTERS=[32, 64], NUM_UPSAMPLE_FILTERS=[64, 64])
    bb = backbones_2d.__all__["BaseBEVBackbone"](cfg, 48)
    bb.load_state_dict(seeded_state_dict(bb, seed=2))
    bb = bb.to(cuda).eval()
    with torch.no_grad():
        out = bb({"spatial_features": torch.from_numpy(g["spatial_features"]).to(cuda)})["spatial_features_2d"].cpu().numpy()
    assert_close_per_channel(out, g["spatial_features_2d"], rtol=1e-3, atol_frac=1e-4, name="spatial_features_2d", channel_axis=1)


@pytest.mark.gpu
@pytest.mark.parametrize("training", [False, True])
def test_hip_bev_backbone_channels_last_with_fused_batchnorm_equals_the_nchw_modules(golden_dir, cuda, hip_lib, training, monkeypatch):
    """BaseBEVBackbone in channels_last (the default from 8 scenes per batch on) with every BatchNorm2d + ReLU on the fused (N H W, C) kernels of the sparse
    backbone, against the same network in NCHW through MIOpen's BatchNorm2d and torch's ReLU: eval against the reference golden, training mode outputs,
    every parameter gradient and the running statistics within 1e-4 of each tensor's scale."""
    import copy
    from seevcn_amd.pcdet.models import backbones_2d
    from seevcn_amd.pcdet.models.backbones_2d import base_bev_backbone as B
    g = np.load(os.path.join(golden_dir, "bev_backbone.npz"))
    cfg = dict(C.SECOND_BACKBONE_2D, LAYER_NUMS=[2, 2], NUM_FILTERS=[32, 64], NUM_UPSAMPLE_FILTERS=[64, 64])
    ref = backbones_2d.__all__["BaseBEVBackbone"](cfg, 48)
    ref.load_state_dict(seeded_state_dict(ref, seed=2))
    ref = ref.to(cuda).train(training)
    new = copy.deepcopy(ref)
    x0 = torch.from_numpy(g["spatial_features"]).to(cuda)
    outs = []
    for net, fmt in ((ref, "nchw"), (new, "nhwc")):
        monkeypatch.setattr(B, "BEV_FORMAT", fmt)
        x = x0.clone().requires_grad_(training)
        with torch.set_grad_enabled(training):
            y = net({"spatial_features": x})["spatial_features_2d"]
            if training:
                (y * torch.linspace(-1, 1, y.numel(), device=cuda).view_as(y)).sum().backward()
        outs.append((y.detach(), x.grad))
    assert outs[1][0].is_contiguous(memory_format=torch.channels_last)
    if not training:
        assert_close_per_channel(outs[1][0].cpu().numpy(), g["spatial_features_2d"], rtol=1e-3, atol_frac=1e-4, name="spatial_features_2d", channel_axis=1)
        return
    def close(a, b, name):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-4 * scale, (name, float((a - b).abs().max()), scale)
    close(outs[1][0], outs[0][0], "output")
    close(outs[1][1], outs[0][1], "input gradient")
    for (k, p), (_, q) in zip(new.named_parameters(), ref.named_parameters()):
        close(p.grad.contiguous(), q.grad.contiguous(), k)
    for (k, p), (_, q) in zip(new.named_buffers(), ref.named_buffers()):
        if p.dtype.is_floating_point:
            close(p, q, k)
        else:
            assert torch.equal(p, q), k


@pytest.mark.gpu
def test_hip_anchor_head_matches_reference_golden(golden_dir, cuda, hip_lib):
    from seevcn_amd.pcdet.models import dense_heads
    g, feat = _golden(golden_dir)
    head = dense_heads.__all__["AnchorHeadSingle"](model_cfg=C.SECOND_DENSE_HEAD, input_channels=16, num_class=3, class_names=C.CLASS_NAMES,
                                                   grid_size=GRID, point_cloud_range=np.array(C.KITTI_RANGE, np.float32))
    head.load_state_dict(seeded_state_dict(head, seed=3))
    head = head.to(cuda).train()
    dd = head({"spatial_features_2d": torch.from_numpy(feat).to(cuda), "gt_boxes": torch.from_numpy(g["gt_boxes"]).to(cuda), "batch_size": 2})
    sel = g["sel"]
    fr = head.forward_ret_dict
    labels = fr["box_cls_labels"].cpu().numpy()
    assert np.array_equal(labels[:, sel], g["box_cls_labels"])                              # integer labels: bit-exact
    assert int((labels > 0).sum()) == int(g["num_pos"]) and int((labels < 0).sum()) == int(g["num_ignore"])
    np.testing.assert_allclose(fr["box_reg_targets"].cpu().numpy()[:, sel], g["box_reg_targets"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(fr["reg_weights"].cpu().numpy()[:, sel], g["reg_weights"])
    np.testing.assert_allclose(dd["batch_cls_preds"].detach().cpu().numpy()[:, sel], g["batch_cls_preds"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(dd["batch_box_preds"].cpu().numpy()[:, sel], g["batch_box_preds"], rtol=1e-3, atol=1e-4)
    loss, tb = head.get_loss()
    for k in ("rpn_loss", "rpn_loss_cls", "rpn_loss_loc", "rpn_loss_dir"):
        assert abs(tb[k] - float(g[k])) <= 1e-3 * abs(float(g[k])) + 1e-5, (k, tb[k], float(g[k]))
    loss.backward()
    assert torch.isfinite(head.conv_box.weight.grad).all()
    # against the oracle on ALL anchors (the golden stores a subsample)
    anchors, per_set = oh.generate_anchors(C.SECOND_DENSE_HEAD["ANCHOR_GENERATOR_CONFIG"], GRID, C.KITTI_RANGE)
    cfgs = C.SECOND_DENSE_HEAD["ANCHOR_GENERATOR_CONFIG"]
    lo, to, wo = oh.assign_targets(anchors, per_set, [1, 2, 3], [c["matched_threshold"] for c in cfgs], [c["unmatched_threshold"] for c in cfgs],
                                   g["gt_boxes"])
    assert np.array_equal(labels, lo)
    np.testing.assert_allclose(fr["box_reg_targets"].cpu().numpy(), to, rtol=1e-4, atol=1e-5)
    # scenes without ground truth -> all background
    head({"spatial_features_2d": torch.from_numpy(feat).to(cuda), "gt_boxes": torch.zeros((2, 3, 8), device=cuda), "batch_size": 2})
    assert int(head.forward_ret_dict["box_cls_labels"].abs().sum()) == 0


@pytest.mark.gpu
def test_hip_second_net_train_and_eval(cuda, hip_lib):
    """SECONDNet built from the registries like the reference's build_network: train step gives finite loss/grads,
    eval post-processing equals a numpy restatement of class_agnostic_nms over the oracle NMS."""
    import seevcn_amd.synth as synth
    from oracle import boxes as ob
    from seevcn_amd.pcdet.models import detectors
    pts, gt = synth.make_scene_batch(2, seed=2000, n_az=100)
    ds = C.SyntheticDatasetInfo()
    net = detectors.build_detector(C.second_model_cfg(), num_class=3, dataset=ds)
    net.load_state_dict(seeded_state_dict(net, seed=4))
    net = net.to(cuda)
    batch = {"batch_size": 2, "points": torch.from_numpy(pts).to(cuda), "gt_boxes": torch.from_numpy(gt).to(cuda)}
    net.train()
    ret, tb, _ = net(dict(batch))
    assert torch.isfinite(ret["loss"]) and set(tb) >= {"loss_rpn", "rpn_loss_cls", "rpn_loss_loc", "rpn_loss_dir", "rpn_loss"}
    ret["loss"].backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    assert net.backbone_3d.conv_input[0].weight.grad is not None and net.backbone_2d.blocks[0][1].weight.grad is not None
    net.eval()
    with torch.no_grad():
        bd = dict(batch)
        for m in net.module_list:
            bd = m(bd)
        pred_dicts, recall = net.post_processing(bd)
    assert len(pred_dicts) == 2 and "gt" in recall
    for i in range(2):
        scores = torch.sigmoid(bd["batch_cls_preds"][i]).max(-1)[0].cpu().numpy()
        boxes = bd["batch_box_preds"][i].cpu().numpy()
        m = np.nonzero(scores >= 0.1)[0]
        order = m[np.argsort(-scores[m], kind="stable")][:4096]
        keep = order[ob.nms(boxes[order], 0.01)][:500]
        got = pred_dicts[i]["pred_boxes"].cpu().numpy()
        assert len(got) == len(keep)
        # identical selection unless two scores tie exactly (sort order of ties is implementation-defined)
        if len(np.unique(scores[m])) == len(m):
            np.testing.assert_allclose(got, boxes[keep], rtol=0, atol=0)


@pytest.mark.gpu
def test_hip_fused_losses_match_the_reference_op_chains(cuda, hip_lib):
    """SigmoidFocalClassificationLoss / WeightedSmoothL1Loss as one launch per direction (sv_sigmoid_focal_loss, sv_weighted_smooth_l1_loss)
    against the reference's chains of torch ops with autograd (loss_utils.py:9-136): values and gradients w.r.t. the predictions."""
    from seevcn_amd.pcdet.utils import loss_utils
    g = torch.Generator().manual_seed(3)
    B, A, C = 3, 5000, 3
    x = (torch.randn(B, A, C, generator=g) * 4.0).to(cuda)
    x[0, :10] = torch.tensor([0.0, 30.0, -30.0])                            # saturated and exactly-zero logits
    t = torch.nn.functional.one_hot(torch.randint(0, C + 1, (B, A), generator=g), C + 1)[..., 1:].float().to(cuda)
    w = torch.rand(B, A, generator=g).to(cuda)
    go = torch.randn(B, A, C, generator=g).to(cuda)
    focal = loss_utils.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0)
    codes = 8
    reg = loss_utils.WeightedSmoothL1Loss(code_weights=[1.0, 1.0, 1.0, 2.0, 0.5, 1.0, 1.0, 3.0])
    p = (torch.randn(B, A, codes, generator=g) * 0.3).to(cuda)
    q = (torch.randn(B, A, codes, generator=g) * 0.3).to(cuda)
    q[1, :50, 2] = float("nan")                                             # ignored targets
    q[2, :50] = p[2, :50]                                                   # zero differences
    go2 = torch.randn(B, A, codes, generator=g).to(cuda)
    res = {}
    for fused in (True, False):
        saved, loss_utils.FUSED_LOSS = loss_utils.FUSED_LOSS, fused
        try:
            xi, pi = x.clone().requires_grad_(True), p.clone().requires_grad_(True)
            lf = focal(xi, t, w)
            lf.backward(go)
            lr = reg(pi, q, w)
            lr.backward(go2)
            lr2 = reg(p, q)                                                 # no anchor weights
            res[fused] = (lf.detach(), xi.grad, lr.detach(), pi.grad, lr2)
        finally:
            loss_utils.FUSED_LOSS = saved
    for name, a, b in zip(("focal", "focal gradient", "smooth-L1", "smooth-L1 gradient", "smooth-L1 without weights"), res[True], res[False]):
        assert a.shape == b.shape and torch.isfinite(a).all(), name
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=1e-6 * float(b.abs().max()), err_msg=name)
    # the point head's call shape: (N, C) logits, (N,) weights
    xi = x[0].clone().requires_grad_(True)
    assert focal(xi, t[0], w[0]).shape == (A, C)
