from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...utils import common_utils, loss_utils
from ...utils.common_utils import cfg_get
from .roi_head_template import RoIHeadTemplate


class SECONDHead(RoIHeadTemplate):
    """IoU-prediction head of SECOND-IoU, the detector SEE-VCN ships weights for (reference roi_heads/second_head.py:10-188):
    proposals (HIP NMS) -> 7x7 rotated RoI grid sampled from the BEV map -> shared FC -> one IoU logit per RoI."""

    def __init__(self, input_channels, model_cfg, num_class=1, **kwargs):
        super().__init__(num_class=num_class, model_cfg=model_cfg)
        self.model_cfg = model_cfg
        pool = cfg_get(model_cfg, 'ROI_GRID_POOL')
        grid = cfg_get(pool, 'GRID_SIZE')
        pre = cfg_get(pool, 'IN_CHANNEL') * grid * grid
        shared, fcs, dp = [], cfg_get(model_cfg, 'SHARED_FC'), cfg_get(model_cfg, 'DP_RATIO')
        for k, c in enumerate(fcs):
            shared += [nn.Conv1d(pre, c, kernel_size=1, bias=False), nn.BatchNorm1d(c), nn.ReLU()]
            pre = c
            if k != len(fcs) - 1 and dp > 0:
                shared.append(nn.Dropout(dp))
        self.shared_fc_layer = nn.Sequential(*shared)
        self.iou_layers = self.make_fc_layers(input_channels=pre, output_channels=1, fc_list=cfg_get(model_cfg, 'IOU_FC'))
        self.init_weights()
        self.affine_grid = partial(F.affine_grid, align_corners=True)
        self.grid_sample = partial(F.grid_sample, align_corners=True)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def roi_grid_pool(self, batch_dict):
        """rois (B,N,7+C), spatial_features_2d (B,C,H,W) -> (B*N, C, G, G): bilinear samples on a GxG grid rotated with the RoI
        (second_head.py:62-116)."""
        batch_size = batch_dict['batch_size']
        rois = batch_dict['rois'].detach()
        feat = batch_dict['spatial_features_2d'].detach()
        height, width = feat.size(2), feat.size(3)
        dcfg = batch_dict['dataset_cfg']
        pc_range = cfg_get(dcfg, 'POINT_CLOUD_RANGE')
        voxel_size = cfg_get(cfg_get(dcfg, 'DATA_PROCESSOR')[-1], 'VOXEL_SIZE')
        min_x, min_y = pc_range[0], pc_range[1]
        pool = cfg_get(self.model_cfg, 'ROI_GRID_POOL')
        ratio, grid_size = cfg_get(pool, 'DOWNSAMPLE_RATIO'), cfg_get(pool, 'GRID_SIZE')
        pooled = []
        for b_id in range(batch_size):
            x1 = (rois[b_id, :, 0] - rois[b_id, :, 3] / 2 - min_x) / (voxel_size[0] * ratio)
            x2 = (rois[b_id, :, 0] + rois[b_id, :, 3] / 2 - min_x) / (voxel_size[0] * ratio)
            y1 = (rois[b_id, :, 1] - rois[b_id, :, 4] / 2 - min_y) / (voxel_size[1] * ratio)
            y2 = (rois[b_id, :, 1] + rois[b_id, :, 4] / 2 - min_y) / (voxel_size[1] * ratio)
            cosa, sina = torch.cos(rois[b_id, :, 6]), torch.sin(rois[b_id, :, 6])
            theta = torch.stack((
                (x2 - x1) / (width - 1) * cosa, (x2 - x1) / (width - 1) * (-sina), (x1 + x2 - width + 1) / (width - 1),
                (y2 - y1) / (height - 1) * sina, (y2 - y1) / (height - 1) * cosa, (y1 + y2 - height + 1) / (height - 1)), dim=1).view(-1, 2, 3).float()
            grid = self.affine_grid(theta, torch.Size((rois.size(1), feat.size(1), grid_size, grid_size)))
            pooled.append(self.grid_sample(feat[b_id].unsqueeze(0).expand(rois.size(1), feat.size(1), height, width), grid))
        return torch.cat(pooled, dim=0)

    def forward(self, batch_dict):
        targets_dict = self.proposal_layer(batch_dict, nms_config=cfg_get(self.model_cfg, 'NMS_CONFIG')['TRAIN' if self.training else 'TEST'])
        if self.training:
            targets_dict = self.assign_targets(batch_dict)
            batch_dict['rois'] = targets_dict['rois']
            batch_dict['roi_labels'] = targets_dict['roi_labels']
        pooled_features = self.roi_grid_pool(batch_dict)
        batch_size_rcnn = pooled_features.shape[0]
        shared_features = self.run_fc(self.shared_fc_layer, pooled_features.reshape(batch_size_rcnn, -1))
        rcnn_iou = self.run_fc(self.iou_layers, shared_features)
        if not self.training:
            batch_dict['batch_cls_preds'] = rcnn_iou.view(batch_dict['batch_size'], -1, rcnn_iou.shape[-1])
            batch_dict['batch_box_preds'] = batch_dict['rois']
            batch_dict['cls_preds_normalized'] = False
        else:
            targets_dict['rcnn_iou'] = rcnn_iou
            self.forward_ret_dict = targets_dict
        return batch_dict

    def get_loss(self, tb_dict=None):
        tb_dict = {} if tb_dict is None else tb_dict
        rcnn_loss, cls_tb_dict = self.get_box_iou_layer_loss(self.forward_ret_dict)
        tb_dict.update(cls_tb_dict)
        tb_dict['rcnn_loss'] = common_utils.tb_value(rcnn_loss)
        return rcnn_loss, tb_dict

    def get_box_iou_layer_loss(self, forward_ret_dict):
        loss_cfgs = cfg_get(self.model_cfg, 'LOSS_CONFIG')
        rcnn_iou_flat = forward_ret_dict['rcnn_iou'].view(-1)
        labels = forward_ret_dict['rcnn_cls_labels'].view(-1)
        kind = cfg_get(loss_cfgs, 'IOU_LOSS')
        if kind == 'BinaryCrossEntropy':
            batch_loss = F.binary_cross_entropy_with_logits(rcnn_iou_flat, labels.float(), reduction='none')
        elif kind == 'L2':
            batch_loss = F.mse_loss(rcnn_iou_flat, labels, reduction='none')
        elif kind == 'smoothL1':
            batch_loss = loss_utils.WeightedSmoothL1Loss.smooth_l1_loss(rcnn_iou_flat - labels, 1.0 / 9.0)
        else:
            raise NotImplementedError
        valid = (labels >= 0).float()
        loss = (batch_loss * valid).sum() / torch.clamp(valid.sum(), min=1.0)
        loss = loss * cfg_get(loss_cfgs, 'LOSS_WEIGHTS')['rcnn_iou_weight']
        return loss, {'rcnn_loss_iou': common_utils.tb_value(loss)}
