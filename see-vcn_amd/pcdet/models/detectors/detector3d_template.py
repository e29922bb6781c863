import torch
import torch.nn as nn

from .. import backbones_2d, backbones_3d, dense_heads, roi_heads
from ..backbones_2d import map_to_bev
from ..backbones_3d import pfe, vfe
from ..model_utils import model_nms_utils
from ...ops.iou3d_nms import iou3d_nms_utils
from ...utils import common_utils
from ...utils.common_utils import cfg_get
from ...utils.spconv_utils import find_all_spconv_keys


class Detector3DTemplate(nn.Module):
    """Composition layer with the reference's contract (detectors/detector3d_template.py:14-173): modules are looked up by
    `model_cfg.<SECTION>.NAME` in the registries, constructed with the same keywords, added under the same attribute names
    (vfe, backbone_3d, map_to_bev_module, backbone_2d, dense_head) and run in `module_topology` order on a batch_dict.
    `dataset` is any object with class_names, grid_size, point_cloud_range, voxel_size, point_feature_encoder.num_point_features."""

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.dataset = dataset
        self.class_names = dataset.class_names
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.module_topology = ['vfe', 'backbone_3d', 'map_to_bev_module', 'pfe', 'backbone_2d', 'dense_head', 'point_head', 'roi_head']

    @property
    def mode(self):
        return 'TRAIN' if self.training else 'TEST'

    def update_global_step(self):
        self.global_step += 1

    def build_networks(self):
        info = {
            'module_list': [],
            'num_rawpoint_features': self.dataset.point_feature_encoder.num_point_features,
            'num_point_features': self.dataset.point_feature_encoder.num_point_features,
            'grid_size': self.dataset.grid_size, 'point_cloud_range': self.dataset.point_cloud_range,
            'voxel_size': self.dataset.voxel_size, 'depth_downsample_factor': getattr(self.dataset, 'depth_downsample_factor', None),
        }
        for name in self.module_topology:
            module, info = getattr(self, 'build_%s' % name)(model_info_dict=info)
            self.add_module(name, module)
        return info['module_list']

    def build_vfe(self, model_info_dict):
        cfg = cfg_get(self.model_cfg, 'VFE', None)
        if cfg is None:
            return None, model_info_dict
        m = vfe.__all__[cfg_get(cfg, 'NAME')](
            model_cfg=cfg, num_point_features=model_info_dict['num_rawpoint_features'], point_cloud_range=model_info_dict['point_cloud_range'],
            voxel_size=model_info_dict['voxel_size'], grid_size=model_info_dict['grid_size'],
            depth_downsample_factor=model_info_dict['depth_downsample_factor'])
        model_info_dict['num_point_features'] = m.get_output_feature_dim()
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def build_backbone_3d(self, model_info_dict):
        cfg = cfg_get(self.model_cfg, 'BACKBONE_3D', None)
        if cfg is None:
            return None, model_info_dict
        m = backbones_3d.__all__[cfg_get(cfg, 'NAME')](
            model_cfg=cfg, input_channels=model_info_dict['num_point_features'], grid_size=model_info_dict['grid_size'],
            voxel_size=model_info_dict['voxel_size'], point_cloud_range=model_info_dict['point_cloud_range'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_point_features'] = m.num_point_features
        model_info_dict['backbone_channels'] = getattr(m, 'backbone_channels', None)
        return m, model_info_dict

    def build_map_to_bev_module(self, model_info_dict):
        cfg = cfg_get(self.model_cfg, 'MAP_TO_BEV', None)
        if cfg is None:
            return None, model_info_dict
        m = map_to_bev.__all__[cfg_get(cfg, 'NAME')](model_cfg=cfg, grid_size=model_info_dict['grid_size'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_bev_features'] = m.num_bev_features
        return m, model_info_dict

    def build_backbone_2d(self, model_info_dict):
        cfg = cfg_get(self.model_cfg, 'BACKBONE_2D', None)
        if cfg is None:
            return None, model_info_dict
        m = backbones_2d.__all__[cfg_get(cfg, 'NAME')](model_cfg=cfg, input_channels=model_info_dict['num_bev_features'])
        for prev in model_info_dict['module_list']:                     # seevcn: HeightCompression may write spatial_features in the backbone's memory format
            if hasattr(prev, 'feeds_bev_backbone') and type(m).__name__ == 'BaseBEVBackbone':
                prev.feeds_bev_backbone = True
        model_info_dict['module_list'].append(m)
        model_info_dict['num_bev_features'] = m.num_bev_features
        return m, model_info_dict

    def build_dense_head(self, model_info_dict):
        cfg = cfg_get(self.model_cfg, 'DENSE_HEAD', None)
        if cfg is None:
            return None, model_info_dict
        m = dense_heads.__all__[cfg_get(cfg, 'NAME')](
            model_cfg=cfg, input_channels=model_info_dict['num_bev_features'],
            num_class=self.num_class if not cfg_get(cfg, 'CLASS_AGNOSTIC', False) else 1, class_names=self.class_names,
            grid_size=model_info_dict['grid_size'], point_cloud_range=model_info_dict['point_cloud_range'],
            predict_boxes_when_training=cfg_get(self.model_cfg, 'ROI_HEAD', None) is not None, voxel_size=model_info_dict.get('voxel_size', False))
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def build_pfe(self, model_info_dict):
        cfg = cfg_get(self.model_cfg, 'PFE', None)
        if cfg is None:
            return None, model_info_dict
        m = pfe.__all__[cfg_get(cfg, 'NAME')](
            model_cfg=cfg, voxel_size=model_info_dict['voxel_size'], point_cloud_range=model_info_dict['point_cloud_range'],
            num_bev_features=model_info_dict['num_bev_features'], num_rawpoint_features=model_info_dict['num_rawpoint_features'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_point_features'] = m.num_point_features
        model_info_dict['num_point_features_before_fusion'] = m.num_point_features_before_fusion
        return m, model_info_dict

    def build_point_head(self, model_info_dict):
        cfg = cfg_get(self.model_cfg, 'POINT_HEAD', None)
        if cfg is None:
            return None, model_info_dict
        if cfg_get(cfg, 'USE_POINT_FEATURES_BEFORE_FUSION', False):
            c = model_info_dict['num_point_features_before_fusion']
        else:
            c = model_info_dict['num_point_features']
        m = dense_heads.__all__[cfg_get(cfg, 'NAME')](
            model_cfg=cfg, input_channels=c, num_class=self.num_class if not cfg_get(cfg, 'CLASS_AGNOSTIC', False) else 1,
            predict_boxes_when_training=cfg_get(self.model_cfg, 'ROI_HEAD', None) is not None)
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def build_roi_head(self, model_info_dict):
        cfg = cfg_get(self.model_cfg, 'ROI_HEAD', None)
        if cfg is None:
            return None, model_info_dict
        m = roi_heads.__all__[cfg_get(cfg, 'NAME')](
            model_cfg=cfg, input_channels=model_info_dict['num_point_features'], backbone_channels=model_info_dict.get('backbone_channels'),
            point_cloud_range=model_info_dict['point_cloud_range'], voxel_size=model_info_dict['voxel_size'],
            num_class=self.num_class if not cfg_get(cfg, 'CLASS_AGNOSTIC', False) else 1)
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    # heads whose get_loss() terms are summed in training, in call order; every head after the first receives the running tb_dict
    LOSS_HEADS = ('dense_head',)

    def run_modules(self, batch_dict):
        modules = list(self.module_list)
        pfe = getattr(self, 'pfe', None)
        if pfe is not None and hasattr(pfe, 'prefetch_keypoints'):
            pfe.prefetch_keypoints(batch_dict)                             # keypoint sampling on a side stream, beside the backbones
            if '_fps_prefetch' in batch_dict:
                # The reference's topology (detector3d_template.py:34-37) puts the pfe in front of backbone_2d and dense_head, but neither reads a
                # key the pfe writes (point_features, point_coords, point_features_before_fusion) nor writes one it reads (points, voxel_coords,
                # multi_scale_3d_features, spatial_features): with the sampling in flight they run first, so that the pfe asks for the keypoints
                # ~10 ms of GPU work after the sampling (7.9 ms) began instead of ~2 ms after.  Same batch_dict, same gradients.
                i = j = modules.index(pfe)
                independent = [m for m in (getattr(self, 'backbone_2d', None), getattr(self, 'dense_head', None)) if m is not None]
                while j + 1 < len(modules) and any(modules[j + 1] is m for m in independent):
                    j += 1
                modules.insert(j, modules.pop(i))
        for module in modules:
            batch_dict = module(batch_dict)
        return batch_dict

    def get_training_loss(self):
        """sum of the LOSS_HEADS' losses -> (loss, tb_dict, disp_dict), the reference detectors' return convention"""
        total, tb_dict = None, None
        with common_utils.deferred_tb():                                   # the heads' loss scalars stay tensors ...
            for name in self.LOSS_HEADS:
                head = getattr(self, name)
                term, tb_dict = head.get_loss() if tb_dict is None else head.get_loss(tb_dict)
                total = term if total is None else total + term
            if len(self.LOSS_HEADS) == 1:
                tb_dict = {'loss_rpn': common_utils.tb_value(total), **tb_dict}
        common_utils.materialize_tb(tb_dict)                               # ... and become Python floats with ONE device -> host read
        return total, tb_dict, {}

    def forward(self, batch_dict):
        """train: ({'loss': loss}, tb_dict, disp_dict); eval: post_processing(batch_dict) = (pred_dicts, recall_dict)"""
        batch_dict = self.run_modules(batch_dict)
        if not self.training:
            return self.post_processing(batch_dict)
        loss, tb_dict, disp_dict = self.get_training_loss()
        return {'loss': loss}, tb_dict, disp_dict

    def post_processing(self, batch_dict):
        """Per scene: sigmoid scores, max over classes, class-agnostic NMS, recall bookkeeping
        (reference detector3d_template.py:178-284, MULTI_CLASSES_NMS False branch)."""
        pp = cfg_get(self.model_cfg, 'POST_PROCESSING')
        nms_cfg = cfg_get(pp, 'NMS_CONFIG')
        assert not cfg_get(nms_cfg, 'MULTI_CLASSES_NMS', False), "multi-class NMS is outside the built path"
        recall_dict, pred_dicts = {}, []
        for index in range(batch_dict['batch_size']):
            if batch_dict.get('batch_index', None) is not None:
                batch_mask = (batch_dict['batch_index'] == index)
            else:
                batch_mask = index
            box_preds = batch_dict['batch_box_preds'][batch_mask]
            src_cls_preds = cls_preds = batch_dict['batch_cls_preds'][batch_mask]
            assert cls_preds.shape[1] in [1, self.num_class]
            if not batch_dict['cls_preds_normalized']:
                cls_preds = torch.sigmoid(cls_preds)
            cls_preds, label_preds = torch.max(cls_preds, dim=-1)
            if batch_dict.get('has_class_labels', False):
                label_key = 'roi_labels' if 'roi_labels' in batch_dict else 'batch_pred_labels'
                label_preds = batch_dict[label_key][index]
            else:
                label_preds = label_preds + 1
            selected, selected_scores = model_nms_utils.class_agnostic_nms(
                box_scores=cls_preds, box_preds=box_preds, nms_config=nms_cfg, score_thresh=cfg_get(pp, 'SCORE_THRESH'))
            if cfg_get(pp, 'OUTPUT_RAW_SCORE', False):
                selected_scores = torch.max(src_cls_preds, dim=-1)[0][selected]
            final_boxes = box_preds[selected]
            recall_dict = self.generate_recall_record(box_preds=final_boxes if 'rois' not in batch_dict else box_preds, recall_dict=recall_dict, batch_index=index, data_dict=batch_dict,
                                                      thresh_list=cfg_get(pp, 'RECALL_THRESH_LIST'))
            pred_dicts.append({'pred_boxes': final_boxes, 'pred_scores': selected_scores, 'pred_labels': label_preds[selected]})
        return pred_dicts, recall_dict

    @staticmethod
    def generate_recall_record(box_preds, recall_dict, batch_index, data_dict=None, thresh_list=None):
        """recall of ground truth at 3-D IoU thresholds (reference :286-328)."""
        if 'gt_boxes' not in data_dict:
            return recall_dict
        rois = data_dict['rois'][batch_index] if 'rois' in data_dict else None
        gt_boxes = data_dict['gt_boxes'][batch_index]
        if len(recall_dict) == 0:
            recall_dict = {'gt': 0}
            for t in thresh_list:
                recall_dict['roi_%s' % str(t)] = 0
                recall_dict['rcnn_%s' % str(t)] = 0
        k = len(gt_boxes) - 1
        while k >= 0 and gt_boxes[k].sum() == 0:
            k -= 1
        cur_gt = gt_boxes[:k + 1]
        if cur_gt.shape[0] > 0:
            iou3d = iou3d_nms_utils.boxes_iou3d_gpu(box_preds[:, 0:7], cur_gt[:, 0:7]) if box_preds.shape[0] > 0 \
                else torch.zeros((0, cur_gt.shape[0]), device=cur_gt.device)
            iou3d_roi = iou3d_nms_utils.boxes_iou3d_gpu(rois[:, 0:7], cur_gt[:, 0:7]) if rois is not None else None
            for t in thresh_list:
                if iou3d.shape[0] > 0:
                    recall_dict['rcnn_%s' % str(t)] += int((iou3d.max(dim=0)[0] > t).sum().item())
                if iou3d_roi is not None:
                    recall_dict['roi_%s' % str(t)] += int((iou3d_roi.max(dim=0)[0] > t).sum().item())
            recall_dict['gt'] += cur_gt.shape[0]
        return recall_dict

    def _load_state_dict(self, model_state_disk, *, strict=True):
        """Checkpoint loading that adapts spconv-1.x weight layouts to this package's 2.x layout (reference :330-359)."""
        state_dict = self.state_dict()
        spconv_keys = find_all_spconv_keys(self)
        update = {}
        for key, val in model_state_disk.items():
            if key in spconv_keys and key in state_dict and state_dict[key].shape != val.shape:
                val_native = val.transpose(-1, -2)              # (k1,k2,k3,c_in,c_out) -> (k1,k2,k3,c_out,c_in)
                if val_native.shape == state_dict[key].shape:
                    val = val_native.contiguous()
                else:
                    assert val.dim() == 5, 'currently only spconv 3D is supported'
                    val_implicit = val.permute(4, 0, 1, 2, 3)   # -> (c_out,k1,k2,k3,c_in)
                    if val_implicit.shape == state_dict[key].shape:
                        val = val_implicit.contiguous()
            if key in state_dict and state_dict[key].shape == val.shape:
                update[key] = val
        if strict:
            self.load_state_dict(update)
        else:
            state_dict.update(update)
            self.load_state_dict(state_dict)
        return state_dict, update
