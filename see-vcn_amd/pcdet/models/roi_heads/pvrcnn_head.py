import torch
import torch.nn as nn

from ...ops.pointnet2.pointnet2_stack import pointnet2_modules as pointnet2_stack_modules
from ...utils import common_utils
from ...utils.common_utils import cfg_get
from .roi_head_template import RoIHeadTemplate


class PVRCNNHead(RoIHeadTemplate):
    """RoI-grid pooling head of PV-RCNN (reference roi_heads/pvrcnn_head.py:8-175): 6^3 grid points per RoI, set abstraction
    of the keypoint features around them (HIP ball query / grouping), shared FC 27648->256->256, cls / reg branches."""

    def __init__(self, input_channels, model_cfg, num_class=1, **kwargs):
        super().__init__(num_class=num_class, model_cfg=model_cfg)
        self.model_cfg = model_cfg
        pool_cfg = cfg_get(model_cfg, 'ROI_GRID_POOL')
        self.roi_grid_pool_layer, num_c_out = pointnet2_stack_modules.build_local_aggregation_module(input_channels=input_channels, config=pool_cfg)
        self.grid_size = cfg_get(pool_cfg, 'GRID_SIZE')
        pre = self.grid_size ** 3 * num_c_out
        shared, fcs, dp = [], cfg_get(model_cfg, 'SHARED_FC'), cfg_get(model_cfg, 'DP_RATIO')
        for k, c in enumerate(fcs):
            shared += [nn.Conv1d(pre, c, kernel_size=1, bias=False), nn.BatchNorm1d(c), nn.ReLU()]
            pre = c
            if k != len(fcs) - 1 and dp > 0:
                shared.append(nn.Dropout(dp))
        self.shared_fc_layer = nn.Sequential(*shared)
        self.cls_layers = self.make_fc_layers(input_channels=pre, output_channels=self.num_class, fc_list=cfg_get(model_cfg, 'CLS_FC'))
        self.reg_layers = self.make_fc_layers(input_channels=pre, output_channels=self.box_coder.code_size * self.num_class,
                                              fc_list=cfg_get(model_cfg, 'REG_FC'))
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.reg_layers[-1].weight, mean=0, std=0.001)

    @staticmethod
    def get_dense_grid_points(rois, batch_size_rcnn, grid_size):
        # rois.new_ones((g, g, g)).nonzero() of the reference = the g^3 index triples in C order: a constant, made once per device
        import numpy as np
        dense_idx = common_utils.const_tensor(np.stack(np.meshgrid(*([np.arange(grid_size)] * 3), indexing='ij'), axis=-1).reshape(-1, 3), rois.device,
                                              rois.dtype).repeat(batch_size_rcnn, 1, 1)                               # (B, g^3, 3)
        size = rois.view(batch_size_rcnn, -1)[:, 3:6]
        return (dense_idx + 0.5) / grid_size * size.unsqueeze(1) - (size.unsqueeze(1) / 2)

    def get_global_grid_points_of_roi(self, rois, grid_size):
        rois = rois.view(-1, rois.shape[-1])
        local = self.get_dense_grid_points(rois, rois.shape[0], grid_size)
        glob = common_utils.rotate_points_along_z(local.clone(), rois[:, 6]).squeeze(dim=1)
        glob = glob + rois[:, 0:3].clone().unsqueeze(dim=1)
        return glob, local

    def roi_grid_pool(self, batch_dict):
        batch_size, rois = batch_dict['batch_size'], batch_dict['rois']
        point_coords = batch_dict['point_coords']
        point_features = batch_dict['point_features'] * batch_dict['point_cls_scores'].view(-1, 1)
        glob, _ = self.get_global_grid_points_of_roi(rois, grid_size=self.grid_size)
        glob = glob.view(batch_size, -1, 3)
        xyz = point_coords[:, 1:4]
        xyz_batch_cnt = common_utils.batch_counts(point_coords[:, 0].long(), batch_size)
        new_xyz = glob.view(-1, 3)
        new_xyz_batch_cnt = xyz.new_zeros(batch_size).int().fill_(glob.shape[1])
        _, pooled = self.roi_grid_pool_layer(xyz=xyz.contiguous(), xyz_batch_cnt=xyz_batch_cnt, new_xyz=new_xyz.contiguous(),
                                             new_xyz_batch_cnt=new_xyz_batch_cnt, features=point_features.contiguous())
        return pooled.view(-1, self.grid_size ** 3, pooled.shape[-1])

    def forward(self, batch_dict):
        nms_cfg = cfg_get(self.model_cfg, 'NMS_CONFIG')['TRAIN' if self.training else 'TEST']
        targets_dict = self.proposal_layer(batch_dict, nms_config=nms_cfg)
        if self.training:
            targets_dict = batch_dict.get('roi_targets_dict', None)
            if targets_dict is None:
                targets_dict = self.assign_targets(batch_dict)
                batch_dict['rois'] = targets_dict['rois']
                batch_dict['roi_labels'] = targets_dict['roi_labels']
        pooled = self.roi_grid_pool(batch_dict)                                   # (BxN, g^3, C)
        n_rcnn = pooled.shape[0]
        pooled = pooled.permute(0, 2, 1).contiguous().view(n_rcnn, -1)            # (BxN, C*g^3)
        shared = self.run_fc(self.shared_fc_layer, pooled)
        rcnn_cls = self.run_fc(self.cls_layers, shared)                           # (BxN, num_class)
        rcnn_reg = self.run_fc(self.reg_layers, shared)                           # (BxN, code_size * num_class)
        if not self.training:
            batch_dict['batch_cls_preds'], batch_dict['batch_box_preds'] = self.generate_predicted_boxes(
                batch_size=batch_dict['batch_size'], rois=batch_dict['rois'], cls_preds=rcnn_cls, box_preds=rcnn_reg)
            batch_dict['cls_preds_normalized'] = False
        else:
            targets_dict['rcnn_cls'], targets_dict['rcnn_reg'] = rcnn_cls, rcnn_reg
            self.forward_ret_dict = targets_dict
        return batch_dict
