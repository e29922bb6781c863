import torch
import torch.nn as nn

from ...ops.roiaware_pool3d import roiaware_pool3d_utils
from ...utils import common_utils, loss_utils
from ...utils.common_utils import cfg_get


class PointHeadTemplate(nn.Module):
    """Point-wise head base with the reference's helpers (dense_heads/point_head_template.py:9-129,131-157)."""

    def __init__(self, model_cfg, num_class):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.add_module('cls_loss_func', loss_utils.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0))
        self.forward_ret_dict = None

    @staticmethod
    def make_fc_layers(fc_cfg, input_channels, output_channels):
        layers, c_in = [], input_channels
        for c in fc_cfg:
            layers += [nn.Linear(c_in, c, bias=False), nn.BatchNorm1d(c), nn.ReLU()]
            c_in = c
        layers.append(nn.Linear(c_in, output_channels, bias=True))
        return nn.Sequential(*layers)

    def assign_stack_targets(self, points, gt_boxes, extend_gt_boxes=None, set_ignore_flag=True, points_per_scene=None):
        """points (N,4) [b,x,y,z] stacked scene by scene with equal counts or ragged; gt_boxes (B,M,8).
        Labels: class (or 1) inside a box, -1 inside the enlarged box only, 0 elsewhere (reference :49-129,
        set_ignore_flag branch).  One batched points-in-boxes launch per box set instead of a python loop over scenes."""
        assert len(points.shape) == 2 and points.shape[1] == 4 and len(gt_boxes.shape) == 3 and gt_boxes.shape[2] == 8
        assert set_ignore_flag, "ball-constraint targets are outside the built path"
        B = gt_boxes.shape[0]
        cnt = common_utils.batch_counts(points[:, 0].long(), B).long()
        # rows of the padded (B, m, 3) block: the per-scene count when the producer of the points stated it (VoxelSetAbstraction: NUM_KEYPOINTS per
        # scene), else an upper bound that needs no device -> host read (all points in one scene)
        m = int(points_per_scene) if points_per_scene else int(points.shape[0])
        padded = points.new_full((B, max(m, 1), 3), 1e8)                     # far-away filler for ragged scenes
        pos = torch.arange(points.shape[0], device=points.device) - (torch.cumsum(cnt, 0) - cnt)[points[:, 0].long()]
        padded[points[:, 0].long(), pos] = points[:, 1:4]
        box_idx = roiaware_pool3d_utils.points_in_boxes_gpu(padded, gt_boxes[:, :, 0:7].contiguous()).long()
        ext_idx = roiaware_pool3d_utils.points_in_boxes_gpu(padded, extend_gt_boxes[:, :, 0:7].contiguous()).long()
        box_idx, ext_idx = box_idx[points[:, 0].long(), pos], ext_idx[points[:, 0].long(), pos]
        fg = box_idx >= 0
        # selects instead of masked assignments (an index list made from a mask is a device -> host read of its length)
        labels = torch.where(fg ^ (ext_idx >= 0), -torch.ones_like(box_idx), torch.zeros_like(box_idx))
        if self.num_class == 1:
            labels = torch.where(fg, torch.ones_like(labels), labels)
        else:
            labels = torch.where(fg, gt_boxes[points[:, 0].long(), box_idx.clamp(min=0), -1].long(), labels)
        return {'point_cls_labels': labels, 'point_box_labels': None, 'point_part_labels': None}

    def get_cls_layer_loss(self, tb_dict=None):
        labels = self.forward_ret_dict['point_cls_labels'].view(-1)
        preds = self.forward_ret_dict['point_cls_preds'].view(-1, self.num_class)
        positives = labels > 0
        cls_weights = ((labels == 0) * 1.0 + 1.0 * positives).float()
        pos_normalizer = positives.sum(dim=0).float()
        cls_weights = cls_weights / torch.clamp(pos_normalizer, min=1.0)
        one_hot = preds.new_zeros(*labels.shape, self.num_class + 1)
        one_hot.scatter_(-1, (labels * (labels >= 0).long()).unsqueeze(-1).long(), 1.0)
        loss = self.cls_loss_func(preds, one_hot[..., 1:], weights=cls_weights).sum()
        loss = loss * cfg_get(self.model_cfg, 'LOSS_CONFIG')['LOSS_WEIGHTS']['point_cls_weight']
        tb_dict = {} if tb_dict is None else tb_dict
        tb_dict.update({'point_loss_cls': common_utils.tb_value(loss), 'point_pos_num': common_utils.tb_value(pos_normalizer)})
        return loss, tb_dict

    def forward(self, **kwargs):
        raise NotImplementedError
