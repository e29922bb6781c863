import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...utils import box_coder_utils, common_utils, loss_utils
from ...utils.common_utils import cfg_get
from ..model_utils.model_nms_utils import class_agnostic_nms, class_agnostic_nms_padded  # noqa: F401
from .target_assigner.proposal_target_layer import ProposalTargetLayer


class RoIHeadTemplate(nn.Module):
    """Second-stage base: proposal NMS, target assignment in the RoI's canonical frame, losses, box decoding
    (reference roi_heads/roi_head_template.py:11-261)."""

    def __init__(self, num_class, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        tcfg = cfg_get(model_cfg, 'TARGET_CONFIG')
        self.box_coder = getattr(box_coder_utils, cfg_get(tcfg, 'BOX_CODER'))(**(cfg_get(tcfg, 'BOX_CODER_CONFIG', {}) or {}))
        self.proposal_target_layer = ProposalTargetLayer(roi_sampler_cfg=tcfg)
        lw = cfg_get(cfg_get(model_cfg, 'LOSS_CONFIG'), 'LOSS_WEIGHTS')
        self.add_module('reg_loss_func', loss_utils.WeightedSmoothL1Loss(code_weights=lw['code_weights']))
        self.forward_ret_dict = None

    def make_fc_layers(self, input_channels, output_channels, fc_list):
        layers, pre = [], input_channels
        dp = cfg_get(self.model_cfg, 'DP_RATIO')
        for k, c in enumerate(fc_list):
            layers += [nn.Conv1d(pre, c, kernel_size=1, bias=False), nn.BatchNorm1d(c), nn.ReLU()]
            pre = c
            if dp >= 0 and k == 0:
                layers.append(nn.Dropout(dp))
        layers.append(nn.Conv1d(pre, output_channels, kernel_size=1, bias=True))
        return nn.Sequential(*layers)

    @staticmethod
    def run_fc(layers, x):
        """The per-RoI FC stacks (Conv1d kernel 1 / BatchNorm1d / ReLU / Dropout over (N, C, 1), pvrcnn_head.py:171-176) on (N, C) rows through the
        library's own dense-layer kernels (seevcn_amd.dense_ops.run_sequential: fp32 MFMA GEMMs with hand-written backward, fused BatchNorm + ReLU).
        Handing MIOpen the (N, C, 1) tensor selects its naive non-packed kernels (40-80 ms per layer at 512 RoIs x 27 648 channels)."""
        from .... import dense_ops
        if x.is_cuda and x.dtype == torch.float32:
            return dense_ops.run_sequential(layers, x)
        for m in layers:
            if isinstance(m, nn.Conv1d):
                assert m.kernel_size == (1,) and m.stride == (1,) and m.padding == (0,) and m.groups == 1
                x = torch.nn.functional.linear(x, m.weight.squeeze(-1), m.bias)
            else:
                x = m(x)
        return x

    @torch.no_grad()
    def proposal_layer(self, batch_dict, nms_config):
        if batch_dict.get('rois', None) is not None:
            return batch_dict
        batch_size = batch_dict['batch_size']
        box_preds_all, cls_preds_all = batch_dict['batch_box_preds'], batch_dict['batch_cls_preds']
        post = cfg_get(nms_config, 'NMS_POST_MAXSIZE')
        rois = box_preds_all.new_zeros((batch_size, post, box_preds_all.shape[-1]))
        roi_scores = box_preds_all.new_zeros((batch_size, post))
        roi_labels = box_preds_all.new_zeros((batch_size, post), dtype=torch.long)
        assert not cfg_get(nms_config, 'MULTI_CLASSES_NMS', False)
        for i in range(batch_size):
            mask = (batch_dict['batch_index'] == i) if batch_dict.get('batch_index', None) is not None else i
            box_preds, cls_preds = box_preds_all[mask], cls_preds_all[mask]
            cur_scores, cur_labels = torch.max(cls_preds, dim=1)
            # the survivors of the NMS in a fixed-size block, the rest zeros: no read of the survivor count on the host
            if box_preds.shape[0] == 0:
                continue                                   # a scene without predictions keeps its zero rows (the reference writes an empty slice)
            selected, valid = class_agnostic_nms_padded(box_scores=cur_scores, box_preds=box_preds, nms_config=nms_config)
            valid = valid.bool()
            # selects, not products: a NaN / inf in the row the padding points at (row 0) must not reach the padded rows
            rois[i] = torch.where(valid.view(-1, 1), box_preds[selected], 0)
            roi_scores[i] = torch.where(valid, cur_scores[selected], 0)
            roi_labels[i] = torch.where(valid, cur_labels[selected], 0)
        batch_dict['rois'], batch_dict['roi_scores'], batch_dict['roi_labels'] = rois, roi_scores, roi_labels + 1
        batch_dict['has_class_labels'] = True if cls_preds_all.shape[-1] > 1 else False
        batch_dict.pop('batch_index', None)
        return batch_dict

    def assign_targets(self, batch_dict):
        batch_size = batch_dict['batch_size']
        with torch.no_grad():
            t = self.proposal_target_layer.forward(batch_dict)
        rois, gt = t['rois'], t['gt_of_rois']
        t['gt_of_rois_src'] = gt.clone().detach()
        roi_center = rois[:, :, 0:3]
        roi_ry = rois[:, :, 6] % (2 * np.pi)
        gt[:, :, 0:3] = gt[:, :, 0:3] - roi_center
        gt[:, :, 6] = gt[:, :, 6] - roi_ry
        gt = common_utils.rotate_points_along_z(points=gt.view(-1, 1, gt.shape[-1]), angle=-roi_ry.view(-1)).view(batch_size, -1, gt.shape[-1])
        heading = gt[:, :, 6] % (2 * np.pi)
        opp = (heading > np.pi * 0.5) & (heading < np.pi * 1.5)
        heading = torch.where(opp, (heading + np.pi) % (2 * np.pi), heading)        # masked assignment as a select: no index list, no host read
        heading = torch.where(heading > np.pi, heading - np.pi * 2, heading)
        gt[:, :, 6] = torch.clamp(heading, min=-np.pi / 2, max=np.pi / 2)
        t['gt_of_rois'] = gt
        return t

    def get_box_reg_layer_loss(self, forward_ret_dict):
        lc = cfg_get(self.model_cfg, 'LOSS_CONFIG')
        lw = cfg_get(lc, 'LOSS_WEIGHTS')
        code = self.box_coder.code_size
        reg_valid_mask = forward_ret_dict['reg_valid_mask'].view(-1)
        gt_ct = forward_ret_dict['gt_of_rois'][..., 0:code]
        gt_src = forward_ret_dict['gt_of_rois_src'][..., 0:code].view(-1, code)
        rcnn_reg, roi_boxes3d = forward_ret_dict['rcnn_reg'], forward_ret_dict['rois']
        n = gt_ct.view(-1, code).shape[0]
        fg_mask = reg_valid_mask > 0
        fg_sum = fg_mask.long().sum()                                              # stays on the device: no host read
        fg_norm = torch.clamp(fg_sum, min=1).float()
        tb = {}
        assert cfg_get(lc, 'REG_LOSS') == 'smooth-l1'
        anchors = roi_boxes3d.clone().detach().view(-1, code)
        anchors[:, 0:3] = 0
        anchors[:, 6] = 0
        reg_targets = self.box_coder.encode_torch(gt_ct.view(n, code), anchors)
        loss = self.reg_loss_func(rcnn_reg.view(n, -1).unsqueeze(0), reg_targets.unsqueeze(0))
        loss = (loss.view(n, -1) * fg_mask.unsqueeze(-1).float()).sum() / fg_norm * lw['rcnn_reg_weight']
        tb['rcnn_loss_reg'] = common_utils.tb_value(loss)
        if cfg_get(lc, 'CORNER_LOSS_REGULARIZATION'):
            # the reference gathers the foreground RoIs (a host read of their number) and averages over them; here every RoI is decoded and the
            # background ones are masked out of the mean: the same value, 0 when there is no foreground RoI
            all_rois = roi_boxes3d.view(1, -1, code)
            batch_anchors = all_rois.clone().detach()
            roi_ry, roi_xyz = all_rois[:, :, 6].view(-1), all_rois[:, :, 0:3].view(-1, 3)
            batch_anchors[:, :, 0:3] = 0
            # mask the INPUTS of the decode, not the product behind it: a background RoI's unsupervised residuals may overflow exp() in
            # decode_torch, and inf * 0 = NaN would poison the loss and every gradient (the reference never decodes a background RoI)
            reg_fg = torch.where(fg_mask.view(1, -1, 1), rcnn_reg.view(1, -1, code), 0)
            boxes = self.box_coder.decode_torch(reg_fg, batch_anchors).view(-1, code)
            boxes = common_utils.rotate_points_along_z(boxes.unsqueeze(1), roi_ry).squeeze(1)
            boxes = torch.cat([boxes[:, 0:3] + roi_xyz, boxes[:, 3:]], dim=1)
            per_roi = loss_utils.get_corner_loss_lidar(boxes[:, 0:7], gt_src[:, 0:7])
            corner = torch.where(fg_mask, per_roi, 0).sum() / fg_norm * lw['rcnn_corner_weight']
            loss = loss + corner
            tb['rcnn_loss_corner'] = common_utils.tb_value(corner)
        return loss, tb

    def get_box_cls_layer_loss(self, forward_ret_dict):
        lc = cfg_get(self.model_cfg, 'LOSS_CONFIG')
        rcnn_cls = forward_ret_dict['rcnn_cls']
        labels = forward_ret_dict['rcnn_cls_labels'].view(-1)
        if cfg_get(lc, 'CLS_LOSS') == 'BinaryCrossEntropy':
            batch_loss = F.binary_cross_entropy(torch.sigmoid(rcnn_cls.view(-1)), labels.float(), reduction='none')
        elif cfg_get(lc, 'CLS_LOSS') == 'CrossEntropy':
            batch_loss = F.cross_entropy(rcnn_cls, labels, reduction='none', ignore_index=-1)
        else:
            raise NotImplementedError
        valid = (labels >= 0).float()
        loss = (batch_loss * valid).sum() / torch.clamp(valid.sum(), min=1.0) * cfg_get(lc, 'LOSS_WEIGHTS')['rcnn_cls_weight']
        return loss, {'rcnn_loss_cls': common_utils.tb_value(loss)}

    def get_loss(self, tb_dict=None):
        tb_dict = {} if tb_dict is None else tb_dict
        cls_loss, tb1 = self.get_box_cls_layer_loss(self.forward_ret_dict)
        reg_loss, tb2 = self.get_box_reg_layer_loss(self.forward_ret_dict)
        tb_dict.update(tb1)
        tb_dict.update(tb2)
        rcnn_loss = cls_loss + reg_loss
        tb_dict['rcnn_loss'] = common_utils.tb_value(rcnn_loss)
        return rcnn_loss, tb_dict

    def generate_predicted_boxes(self, batch_size, rois, cls_preds, box_preds):
        code = self.box_coder.code_size
        batch_cls_preds = cls_preds.view(batch_size, -1, cls_preds.shape[-1])
        roi_ry, roi_xyz = rois[:, :, 6].view(-1), rois[:, :, 0:3].view(-1, 3)
        local_rois = rois.clone().detach()
        local_rois[:, :, 0:3] = 0
        boxes = self.box_coder.decode_torch(box_preds.view(batch_size, -1, code), local_rois).view(-1, code)
        boxes = common_utils.rotate_points_along_z(boxes.unsqueeze(1), roi_ry).squeeze(1)
        boxes[:, 0:3] += roi_xyz
        return batch_cls_preds, boxes.view(batch_size, -1, code)
