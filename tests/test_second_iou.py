"""SECOND-IoU (the detector SEE-VCN ships weights for): SECONDHead (proposals via HIP NMS, rotated RoI grid
sampling of the BEV map, IoU branch, loss) and SECONDNetIoU.post_processing against the reference's own modules run on CPU
(tests/golden/make_second_iou_golden.py)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from pvrcnn_inputs import make_inputs
from second_iou_inputs import DATASET_CFG, HEAD_KW, bev_map
from seeding import seeded_state_dict
from tolerances import assert_close_per_channel
from seevcn_amd.pcdet import model_cfgs as C


def _head():
    from seevcn_amd.pcdet.models import roi_heads
    cfg = C.second_iou_roi_head(**HEAD_KW)
    cfg['DP_RATIO'] = 0.0
    head = roi_heads.__all__['SECONDHead'](input_channels=8, model_cfg=cfg, num_class=1)
    head.load_state_dict(seeded_state_dict(head, seed=21))
    return head


def test_second_head_state_dict_keys():
    sd = _head().state_dict()
    assert sd['shared_fc_layer.0.weight'].shape == (64, 8 * 7 * 7, 1) and 'iou_layers.7.weight' in sd
    from seevcn_amd.pcdet.models import detectors
    assert 'SECONDNetIoU' in detectors.__all__


@pytest.mark.gpu
def test_hip_second_iou_head_and_post_processing_match_reference_golden(golden_dir, cuda, hip_lib):
    from seevcn_amd.pcdet.models.detectors.detector3d_template import Detector3DTemplate
    from seevcn_amd.pcdet.models.detectors.second_net_iou import SECONDNetIoU
    g = np.load(os.path.join(golden_dir, "second_iou.npz"))
    inp = make_inputs()
    head = _head().to(cuda)
    t = lambda a: torch.from_numpy(a).to(cuda)

    def batch():
        return {'batch_size': 2, 'gt_boxes': t(inp['gt_boxes']), 'spatial_features_2d': t(bev_map()), 'batch_cls_preds': t(inp['batch_cls_preds']),
                'batch_box_preds': t(inp['batch_box_preds']), 'cls_preds_normalized': False, 'dataset_cfg': DATASET_CFG}

    head.train()
    np.random.seed(7)
    torch.manual_seed(7)
    head(batch())
    fr = head.forward_ret_dict
    np.testing.assert_allclose(fr['rois'].cpu().numpy(), g['train_rois'], rtol=0, atol=0)            # same NMS survivors, same random sample
    np.testing.assert_allclose(fr['rcnn_cls_labels'].cpu().numpy(), g['rcnn_cls_labels'], rtol=1e-3, atol=1e-4)
    assert_close_per_channel(fr['rcnn_iou'].detach().cpu().numpy(), g['rcnn_iou'], rtol=1e-3, atol_frac=1e-4, name="rcnn_iou")
    loss, tb = head.get_loss()
    assert abs(tb['rcnn_loss_iou'] - float(g['rcnn_loss_iou'])) <= 1e-3 * float(g['rcnn_loss_iou'])
    loss.backward()
    assert torch.isfinite(head.shared_fc_layer[0].weight.grad).all()
    head.eval()
    with torch.no_grad():
        bd = head(batch())
    np.testing.assert_allclose(bd['rois'].cpu().numpy(), g['eval_rois'], rtol=0, atol=0)
    assert np.array_equal(bd['roi_labels'].cpu().numpy(), g['eval_roi_labels'])
    np.testing.assert_allclose(bd['roi_scores'].cpu().numpy(), g['eval_roi_scores'], rtol=1e-6, atol=0)
    assert_close_per_channel(bd['batch_cls_preds'].cpu().numpy(), g['eval_batch_cls_preds'], rtol=1e-3, atol_frac=1e-4, name="eval_batch_cls_preds")
    fake = SimpleNamespace(model_cfg=dict(POST_PROCESSING=C.SECOND_IOU_POST), num_class=3, class_names=C.CLASS_NAMES,
                           generate_recall_record=Detector3DTemplate.generate_recall_record)
    with torch.no_grad():
        preds, recall = SECONDNetIoU.post_processing(fake, bd)
    for k, p in enumerate(preds):
        # NMS at 0.01 on IoU-logit scores that differ in the last bits between CPU and GPU convs: compare as sets
        gb, gs = g[f'pred_boxes_{k}'], g[f'pred_scores_{k}']
        pb, ps = p['pred_boxes'].cpu().numpy(), p['pred_scores'].cpu().numpy()
        assert abs(len(pb) - len(gb)) <= 1
        hit = sum((np.abs(pb - gb[i]).max(1) + np.abs(ps - gs[i])).min() < 2e-3 for i in range(len(gb)))
        assert hit >= len(gb) - 1
        assert np.array_equal(np.sort(p['pred_labels'].cpu().numpy()), np.sort(g[f'pred_labels_{k}'])) or abs(len(pb) - len(gb)) == 1
    for k, v in recall.items():
        assert v == float(g['recall_' + k]), k
