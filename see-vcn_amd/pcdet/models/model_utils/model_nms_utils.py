import torch

from ...ops.iou3d_nms import iou3d_nms_utils
from ...utils.common_utils import cfg_get


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """score threshold -> top-k (NMS_PRE_MAXSIZE) -> NMS_TYPE(NMS_THRESH) -> first NMS_POST_MAXSIZE; returns indices into the
    original arrays and their scores (reference model_utils/model_nms_utils.py:6-25)."""
    src_box_scores = box_scores
    if score_thresh is not None:
        scores_mask = (box_scores >= score_thresh)
        box_scores = box_scores[scores_mask]
        box_preds = box_preds[scores_mask]
    selected = []
    if box_scores.shape[0] > 0:
        box_scores_nms, indices = torch.topk(box_scores, k=min(cfg_get(nms_config, 'NMS_PRE_MAXSIZE'), box_scores.shape[0]))
        boxes_for_nms = box_preds[indices]
        keep_idx, _ = getattr(iou3d_nms_utils, cfg_get(nms_config, 'NMS_TYPE'))(boxes_for_nms[:, 0:7], box_scores_nms,
                                                                                 cfg_get(nms_config, 'NMS_THRESH'),
                                                                                 max_keep=cfg_get(nms_config, 'NMS_POST_MAXSIZE'))
        selected = indices[keep_idx[:cfg_get(nms_config, 'NMS_POST_MAXSIZE')]]
    if score_thresh is not None:
        original_idxs = scores_mask.nonzero().view(-1)
        selected = original_idxs[selected]
    return selected, src_box_scores[selected]


def class_agnostic_nms_padded(box_scores, box_preds, nms_config):
    """class_agnostic_nms without a score threshold as a fixed-size result: (selected (NMS_POST_MAXSIZE,) indices into the original arrays,
    valid (NMS_POST_MAXSIZE,) bool); the survivors come first, in score order.  No device -> host read."""
    post = cfg_get(nms_config, 'NMS_POST_MAXSIZE')
    if box_scores.shape[0] == 0:
        z = torch.zeros((post,), dtype=torch.int64, device=box_scores.device)
        return z, z.bool()
    box_scores_nms, indices = torch.topk(box_scores, k=min(cfg_get(nms_config, 'NMS_PRE_MAXSIZE'), box_scores.shape[0]))
    # torch.topk returns its values in descending order: the NMS need not sort them again
    keep, valid = iou3d_nms_utils.nms_gpu_padded(box_preds[indices][:, 0:7], box_scores_nms, cfg_get(nms_config, 'NMS_THRESH'), post,
                                                 normal=cfg_get(nms_config, 'NMS_TYPE') == 'nms_normal_gpu', presorted=True)
    return indices[keep], valid
