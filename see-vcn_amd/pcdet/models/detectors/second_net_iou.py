import torch

from ...utils.common_utils import cfg_get
from ..model_utils.model_nms_utils import class_agnostic_nms
from .detector3d_template import Detector3DTemplate


class SECONDNetIoU(Detector3DTemplate):
    """SECOND with the IoU-rectification head (reference detectors/second_net_iou.py:7-177), the model behind the SEE-VCN demo
    weights.  post_processing combines the RPN class score and the predicted IoU per SCORE_TYPE before the (HIP) NMS."""

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    LOSS_HEADS = ('dense_head', 'roi_head')

    def forward(self, batch_dict):
        batch_dict['dataset_cfg'] = self.dataset.dataset_cfg
        return super().forward(batch_dict)

    @staticmethod
    def cal_scores_by_npoints(cls_scores, iou_scores, num_points_in_gt, cls_thresh=10, iou_thresh=100):
        assert iou_thresh >= cls_thresh
        alpha = torch.zeros(cls_scores.shape, dtype=torch.float32, device=cls_scores.device)
        alpha[num_points_in_gt >= iou_thresh] = 1
        mask = (num_points_in_gt > cls_thresh) & (num_points_in_gt < iou_thresh)
        alpha[mask] = (num_points_in_gt[mask] - 10) / (iou_thresh - cls_thresh)
        return (1 - alpha) * cls_scores + alpha * iou_scores

    def set_nms_score_by_class(self, iou_preds, cls_preds, label_preds, score_by_class):
        n_classes = torch.unique(label_preds).shape[0]
        nms_scores = torch.zeros(iou_preds.shape, dtype=torch.float32, device=iou_preds.device)
        for i in range(n_classes):
            mask = label_preds == (i + 1)
            score_type = score_by_class[self.class_names[i]]
            if score_type == 'iou':
                nms_scores[mask] = iou_preds[mask]
            elif score_type == 'cls':
                nms_scores[mask] = cls_preds[mask]
            else:
                raise NotImplementedError
        return nms_scores

    def post_processing(self, batch_dict):
        pp = cfg_get(self.model_cfg, 'POST_PROCESSING')
        nms_cfg = cfg_get(pp, 'NMS_CONFIG')
        recall_dict, pred_dicts = {}, []
        for index in range(batch_dict['batch_size']):
            if batch_dict.get('batch_index', None) is not None:
                assert batch_dict['batch_cls_preds'].dim() == 2
                batch_mask = batch_dict['batch_index'] == index
            else:
                assert batch_dict['batch_cls_preds'].dim() == 3
                batch_mask = index
            box_preds = batch_dict['batch_box_preds'][batch_mask]
            iou_preds = batch_dict['batch_cls_preds'][batch_mask]
            cls_preds = batch_dict['roi_scores'][batch_mask]
            src_box_preds = box_preds
            assert iou_preds.shape[1] in [1, self.num_class]
            if not batch_dict['cls_preds_normalized']:
                iou_preds, cls_preds = torch.sigmoid(iou_preds), torch.sigmoid(cls_preds)
            if cfg_get(nms_cfg, 'MULTI_CLASSES_NMS'):
                raise NotImplementedError
            iou_preds, label_preds = torch.max(iou_preds, dim=-1)
            label_preds = batch_dict['roi_labels'][index] if batch_dict.get('has_class_labels', False) else label_preds + 1
            score_type = cfg_get(nms_cfg, 'SCORE_TYPE', None)
            if cfg_get(nms_cfg, 'SCORE_BY_CLASS', None) and score_type == 'score_by_class':
                nms_scores = self.set_nms_score_by_class(iou_preds, cls_preds, label_preds, cfg_get(nms_cfg, 'SCORE_BY_CLASS'))
            elif score_type == 'iou' or score_type is None:
                nms_scores = iou_preds
            elif score_type == 'cls':
                nms_scores = cls_preds
            elif score_type == 'weighted_iou_cls':
                w = cfg_get(nms_cfg, 'SCORE_WEIGHTS')
                nms_scores = cfg_get(w, 'iou') * iou_preds + cfg_get(w, 'cls') * cls_preds
            elif score_type == 'num_pts_iou_cls':
                from ...ops.roiaware_pool3d import roiaware_pool3d_utils
                pts = batch_dict['points']
                batch_points = pts[pts[:, 0] == batch_mask][:, 1:4]
                # reference: points_in_boxes_cpu(points, boxes).sum(1) -> points per box; here one GPU launch + bincount
                inside = roiaware_pool3d_utils.points_in_boxes_gpu(batch_points.unsqueeze(0), box_preds[:, 0:7].unsqueeze(0)).view(-1)
                num_pts = torch.bincount(inside[inside >= 0].long(), minlength=box_preds.shape[0]).float()
                th = cfg_get(nms_cfg, 'SCORE_THRESH')
                nms_scores = self.cal_scores_by_npoints(cls_preds, iou_preds, num_pts, cfg_get(th, 'cls'), cfg_get(th, 'iou'))
            else:
                raise NotImplementedError
            selected, selected_scores = class_agnostic_nms(box_scores=nms_scores, box_preds=box_preds, nms_config=nms_cfg,
                                                           score_thresh=cfg_get(pp, 'SCORE_THRESH'))
            if cfg_get(pp, 'OUTPUT_RAW_SCORE'):
                raise NotImplementedError
            final_boxes = box_preds[selected]
            recall_dict = self.generate_recall_record(box_preds=final_boxes if 'rois' not in batch_dict else src_box_preds, recall_dict=recall_dict,
                                                      batch_index=index, data_dict=batch_dict, thresh_list=cfg_get(pp, 'RECALL_THRESH_LIST'))
            pred_dicts.append({'pred_boxes': final_boxes, 'pred_scores': selected_scores, 'pred_labels': label_preds[selected],
                               'pred_cls_scores': cls_preds[selected], 'pred_iou_scores': iou_preds[selected]})
        return pred_dicts, recall_dict
