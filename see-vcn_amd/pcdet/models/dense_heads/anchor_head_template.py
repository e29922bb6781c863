import numpy as np
import torch
import torch.nn as nn

from .... import _lib
from ...utils import box_coder_utils, common_utils, loss_utils
from ...utils.common_utils import cfg_get
from .target_assigner.anchor_generator import AnchorGenerator
from .target_assigner.axis_aligned_target_assigner import AxisAlignedTargetAssigner


class AnchorHeadTemplate(nn.Module):
    """Same constructor, attributes (`anchors`, `box_coder`, `forward_ret_dict`), loss names and tb_dict keys as the reference
    AnchorHeadTemplate (dense_heads/anchor_head_template.py:11-275). Anchors are created on the first use of a device."""

    def __init__(self, model_cfg, num_class, class_names, grid_size, point_cloud_range, predict_boxes_when_training):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.class_names = class_names
        self.predict_boxes_when_training = predict_boxes_when_training
        self.use_multihead = cfg_get(model_cfg, 'USE_MULTIHEAD', False)
        assert not self.use_multihead, "AnchorHeadMulti is outside the built path"
        anchor_target_cfg = cfg_get(model_cfg, 'TARGET_ASSIGNER_CONFIG')
        self.box_coder = getattr(box_coder_utils, cfg_get(anchor_target_cfg, 'BOX_CODER'))(
            num_dir_bins=cfg_get(anchor_target_cfg, 'NUM_DIR_BINS', 6), **(cfg_get(anchor_target_cfg, 'BOX_CODER_CONFIG', {}) or {}))
        anchor_generator_cfg = cfg_get(model_cfg, 'ANCHOR_GENERATOR_CONFIG')
        anchors, self.num_anchors_per_location = self.generate_anchors(
            anchor_generator_cfg, grid_size=grid_size, point_cloud_range=point_cloud_range, anchor_ndim=self.box_coder.code_size)
        self.anchors = anchors
        self.target_assigner = self.get_target_assigner(anchor_target_cfg)
        self.forward_ret_dict = {}
        self.build_losses(cfg_get(model_cfg, 'LOSS_CONFIG'))
        self._flat_anchors = None

    @staticmethod
    def generate_anchors(anchor_generator_cfg, grid_size, point_cloud_range, anchor_ndim=7):
        gen = AnchorGenerator(anchor_range=point_cloud_range, anchor_generator_config=anchor_generator_cfg)
        grid_size = np.asarray(grid_size)
        feature_map_size = [grid_size[:2] // c['feature_map_stride'] for c in anchor_generator_cfg]
        anchors_list, num_per_loc = gen.generate_anchors(feature_map_size)
        if anchor_ndim != 7:
            anchors_list = [torch.cat((a, a.new_zeros([*a.shape[0:-1], anchor_ndim - 7])), dim=-1) for a in anchors_list]
        return anchors_list, num_per_loc

    def get_target_assigner(self, anchor_target_cfg):
        if cfg_get(anchor_target_cfg, 'NAME') == 'AxisAlignedTargetAssigner':
            return AxisAlignedTargetAssigner(model_cfg=self.model_cfg, class_names=self.class_names, box_coder=self.box_coder,
                                             match_height=cfg_get(anchor_target_cfg, 'MATCH_HEIGHT', False))
        raise NotImplementedError

    def build_losses(self, losses_cfg):
        self.add_module('cls_loss_func', loss_utils.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0))
        reg_loss_name = cfg_get(losses_cfg, 'REG_LOSS_TYPE', None) or 'WeightedSmoothL1Loss'
        self.add_module('reg_loss_func', getattr(loss_utils, reg_loss_name)(code_weights=cfg_get(losses_cfg, 'LOSS_WEIGHTS')['code_weights']))
        self.add_module('dir_loss_func', loss_utils.WeightedCrossEntropyLoss())

    def _anchors_on(self, device):
        if self.anchors[0].device != device:
            self.anchors = [a.to(device) for a in self.anchors]
            self._flat_anchors = None
        if self._flat_anchors is None:
            self._flat_anchors = torch.cat(self.anchors, dim=-3).reshape(-1, self.anchors[0].shape[-1]).contiguous()
        return self._flat_anchors

    def assign_targets(self, gt_boxes):
        self._anchors_on(gt_boxes.device)
        return self.target_assigner.assign_targets(self.anchors, gt_boxes)

    def get_cls_layer_loss(self):
        cls_preds = self.forward_ret_dict['cls_preds']
        box_cls_labels = self.forward_ret_dict['box_cls_labels']
        batch_size = int(cls_preds.shape[0])
        cared = box_cls_labels >= 0
        positives = box_cls_labels > 0
        negatives = box_cls_labels == 0
        cls_weights = (negatives * 1.0 + 1.0 * positives).float()
        if self.num_class == 1:
            box_cls_labels = torch.where(positives, torch.ones_like(box_cls_labels), box_cls_labels)  # class agnostic
        pos_normalizer = positives.sum(1, keepdim=True).float()
        cls_weights = cls_weights / torch.clamp(pos_normalizer, min=1.0)
        cls_targets = (box_cls_labels * cared.type_as(box_cls_labels)).long()
        one_hot = torch.zeros(*cls_targets.shape, self.num_class + 1, dtype=cls_preds.dtype, device=cls_targets.device)
        one_hot.scatter_(-1, cls_targets.unsqueeze(-1), 1.0)
        cls_loss_src = self.cls_loss_func(cls_preds.view(batch_size, -1, self.num_class), one_hot[..., 1:], weights=cls_weights)
        cls_loss = cls_loss_src.sum() / batch_size * cfg_get(self.model_cfg, 'LOSS_CONFIG')['LOSS_WEIGHTS']['cls_weight']
        return cls_loss, {'rpn_loss_cls': common_utils.tb_value(cls_loss)}

    @staticmethod
    def add_sin_difference(boxes1, boxes2, dim=6):
        assert dim != -1
        enc1 = torch.sin(boxes1[..., dim:dim + 1]) * torch.cos(boxes2[..., dim:dim + 1])
        enc2 = torch.cos(boxes1[..., dim:dim + 1]) * torch.sin(boxes2[..., dim:dim + 1])
        return (torch.cat([boxes1[..., :dim], enc1, boxes1[..., dim + 1:]], dim=-1),
                torch.cat([boxes2[..., :dim], enc2, boxes2[..., dim + 1:]], dim=-1))

    @staticmethod
    def get_direction_target(anchors, reg_targets, one_hot=True, dir_offset=0, num_bins=2):
        from ...utils import common_utils
        batch_size = reg_targets.shape[0]
        anchors = anchors.view(batch_size, -1, anchors.shape[-1])
        rot_gt = reg_targets[..., 6] + anchors[..., 6]
        offset_rot = common_utils.limit_period(rot_gt - dir_offset, 0, 2 * np.pi)
        dir_cls_targets = torch.clamp(torch.floor(offset_rot / (2 * np.pi / num_bins)).long(), min=0, max=num_bins - 1)
        if one_hot:
            dir_targets = torch.zeros(*dir_cls_targets.shape, num_bins, dtype=anchors.dtype, device=dir_cls_targets.device)
            dir_targets.scatter_(-1, dir_cls_targets.unsqueeze(-1), 1.0)
            dir_cls_targets = dir_targets
        return dir_cls_targets

    def get_box_reg_layer_loss(self):
        box_preds = self.forward_ret_dict['box_preds']
        box_dir_cls_preds = self.forward_ret_dict.get('dir_cls_preds', None)
        box_reg_targets = self.forward_ret_dict['box_reg_targets']
        box_cls_labels = self.forward_ret_dict['box_cls_labels']
        batch_size = int(box_preds.shape[0])
        positives = box_cls_labels > 0
        reg_weights = positives.float()
        reg_weights = reg_weights / torch.clamp(positives.sum(1, keepdim=True).float(), min=1.0)
        anchors = self._anchors_on(box_preds.device).view(1, -1, self.anchors[0].shape[-1]).repeat(batch_size, 1, 1)
        box_preds = box_preds.view(batch_size, -1, box_preds.shape[-1] // self.num_anchors_per_location)
        box_preds_sin, reg_targets_sin = self.add_sin_difference(box_preds, box_reg_targets)
        loss_w = cfg_get(self.model_cfg, 'LOSS_CONFIG')['LOSS_WEIGHTS']
        loc_loss = self.reg_loss_func(box_preds_sin, reg_targets_sin, weights=reg_weights).sum() / batch_size * loss_w['loc_weight']
        box_loss = loc_loss
        tb_dict = {'rpn_loss_loc': common_utils.tb_value(loc_loss)}
        if box_dir_cls_preds is not None:
            nb = cfg_get(self.model_cfg, 'NUM_DIR_BINS')
            dir_targets = self.get_direction_target(anchors, box_reg_targets, dir_offset=cfg_get(self.model_cfg, 'DIR_OFFSET'), num_bins=nb)
            dir_logits = box_dir_cls_preds.view(batch_size, -1, nb)
            weights = positives.type_as(dir_logits)
            weights = weights / torch.clamp(weights.sum(-1, keepdim=True), min=1.0)
            dir_loss = self.dir_loss_func(dir_logits, dir_targets, weights=weights).sum() / batch_size * loss_w['dir_weight']
            box_loss = box_loss + dir_loss
            tb_dict['rpn_loss_dir'] = common_utils.tb_value(dir_loss)
        return box_loss, tb_dict

    def get_loss(self):
        cls_loss, tb_dict = self.get_cls_layer_loss()
        box_loss, tb_dict_box = self.get_box_reg_layer_loss()
        tb_dict.update(tb_dict_box)
        rpn_loss = cls_loss + box_loss
        tb_dict['rpn_loss'] = common_utils.tb_value(rpn_loss)
        return rpn_loss, tb_dict

    @torch.no_grad()
    def generate_predicted_boxes(self, batch_size, cls_preds, box_preds, dir_cls_preds=None):
        """cls_preds (N,H,W,C1), box_preds (N,H,W,C2), dir_cls_preds (N,H,W,C3) -> (B,num_boxes,num_classes), (B,num_boxes,7)
        through the fused decode kernel sv_anchor_decode."""
        lib = _lib.load()
        _lib.require_cuda(box_preds)
        anchors = self._anchors_on(box_preds.device)
        assert anchors.shape[-1] == 7, "fused decode handles 7-d boxes"
        num_anchors = anchors.shape[0]
        batch_cls_preds = cls_preds.view(batch_size, num_anchors, -1).float()
        enc = box_preds.reshape(batch_size, num_anchors, -1).contiguous().float()
        out = torch.empty_like(enc)
        dirp = nb = None
        if dir_cls_preds is not None:
            nb = cfg_get(self.model_cfg, 'NUM_DIR_BINS')
            dirp = dir_cls_preds.reshape(batch_size, num_anchors, -1).contiguous().float()
        rc = lib.sv_anchor_decode(_lib.ptr(anchors), num_anchors, _lib.ptr(enc), _lib.ptr(dirp), batch_size, int(nb or 0),
                                  float(cfg_get(self.model_cfg, 'DIR_OFFSET', 0.0)), float(cfg_get(self.model_cfg, 'DIR_LIMIT_OFFSET', 0.0)),
                                  _lib.ptr(out), _lib.stream())
        _lib.check(rc, "sv_anchor_decode")
        return batch_cls_preds, out

    def forward(self, **kwargs):
        raise NotImplementedError
