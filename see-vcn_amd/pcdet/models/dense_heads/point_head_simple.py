import torch

from ...utils import box_utils
from ...utils.common_utils import cfg_get
from .point_head_template import PointHeadTemplate


class PointHeadSimple(PointHeadTemplate):
    """Keypoint foreground segmentation head of PV-RCNN (reference dense_heads/point_head_simple.py:7-91)."""

    def __init__(self, num_class, input_channels, model_cfg, **kwargs):
        super().__init__(model_cfg=model_cfg, num_class=num_class)
        self.cls_layers = self.make_fc_layers(fc_cfg=cfg_get(model_cfg, 'CLS_FC'), input_channels=input_channels, output_channels=num_class)

    def assign_targets(self, input_dict):
        point_coords, gt_boxes = input_dict['point_coords'], input_dict['gt_boxes']
        assert gt_boxes.dim() == 3 and point_coords.dim() == 2
        batch_size = gt_boxes.shape[0]
        extend = box_utils.enlarge_box3d(gt_boxes.view(-1, gt_boxes.shape[-1]),
                                         extra_width=cfg_get(self.model_cfg, 'TARGET_CONFIG')['GT_EXTRA_WIDTH']).view(batch_size, -1, gt_boxes.shape[-1])
        return self.assign_stack_targets(points=point_coords, gt_boxes=gt_boxes, extend_gt_boxes=extend, set_ignore_flag=True,
                                         points_per_scene=input_dict.get('point_coords_per_scene'))

    def get_loss(self, tb_dict=None):
        tb_dict = {} if tb_dict is None else tb_dict
        loss, tb1 = self.get_cls_layer_loss()
        tb_dict.update(tb1)
        return loss, tb_dict

    def forward(self, batch_dict):
        key = 'point_features_before_fusion' if cfg_get(self.model_cfg, 'USE_POINT_FEATURES_BEFORE_FUSION', False) else 'point_features'
        from .... import dense_ops
        point_cls_preds = dense_ops.run_sequential(self.cls_layers, batch_dict[key])       # own GEMM / BatchNorm kernels (module tree for CPU tensors)
        ret = {'point_cls_preds': point_cls_preds}
        batch_dict['point_cls_scores'], _ = torch.sigmoid(point_cls_preds).max(dim=-1)
        if self.training:
            ret['point_cls_labels'] = self.assign_targets(batch_dict)['point_cls_labels']
        self.forward_ret_dict = ret
        return batch_dict
