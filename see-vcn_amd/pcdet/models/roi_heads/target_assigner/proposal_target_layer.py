import numpy as np
import torch
import torch.nn as nn

from ....ops.iou3d_nms import iou3d_nms_utils
from ....utils.common_utils import cfg_get


class ProposalTargetLayer(nn.Module):
    """RoI sampling + labels for the second stage (reference roi_heads/target_assigner/proposal_target_layer.py:8-230).
    The random draws consume numpy's global RNG and torch's CPU generator in the same order as the reference
    (np.random.permutation for foreground, torch.randint for hard then easy background), so a seeded run selects the same RoIs."""

    def __init__(self, roi_sampler_cfg):
        super().__init__()
        self.roi_sampler_cfg = roi_sampler_cfg

    def _c(self, key, default=None):
        return cfg_get(self.roi_sampler_cfg, key, default)

    def forward(self, batch_dict):
        rois, gt_of_rois, ious, scores, labels = self.sample_rois_for_rcnn(batch_dict)
        reg_valid_mask = (ious > self._c('REG_FG_THRESH')).long()
        st = self._c('CLS_SCORE_TYPE')
        if st == 'cls':
            cls_labels = (ious > self._c('CLS_FG_THRESH')).long()
            cls_labels[(ious > self._c('CLS_BG_THRESH')) & (ious < self._c('CLS_FG_THRESH'))] = -1
        elif st == 'roi_iou':
            bg, fg = self._c('CLS_BG_THRESH'), self._c('CLS_FG_THRESH')
            fg_mask, bg_mask = ious > fg, ious < bg
            interval = (fg_mask == 0) & (bg_mask == 0)
            cls_labels = (fg_mask > 0).float()
            cls_labels[interval] = (ious[interval] - bg) / (fg - bg)
        elif st == 'raw_roi_iou':
            cls_labels = ious
        else:
            raise NotImplementedError
        return {'rois': rois, 'gt_of_rois': gt_of_rois, 'gt_iou_of_rois': ious, 'roi_scores': scores, 'roi_labels': labels,
                'reg_valid_mask': reg_valid_mask, 'rcnn_cls_labels': cls_labels}

    def sample_rois_for_rcnn(self, batch_dict):
        batch_size = batch_dict['batch_size']
        rois, roi_scores, roi_labels, gt_boxes = batch_dict['rois'], batch_dict['roi_scores'], batch_dict['roi_labels'], batch_dict['gt_boxes']
        n, code = self._c('ROI_PER_IMAGE'), rois.shape[-1]
        b_rois = rois.new_zeros(batch_size, n, code)
        b_gt = rois.new_zeros(batch_size, n, code + 1)
        b_iou = rois.new_zeros(batch_size, n)
        b_scores = rois.new_zeros(batch_size, n)
        b_labels = rois.new_zeros((batch_size, n), dtype=torch.long)
        for i in range(batch_size):
            cur_roi, cur_gt, cur_labels, cur_scores = rois[i], gt_boxes[i], roi_labels[i], roi_scores[i]
            k = len(cur_gt) - 1
            while k >= 0 and cur_gt[k].sum() == 0:
                k -= 1
            cur_gt = cur_gt[:k + 1]
            cur_gt = cur_gt.new_zeros((1, cur_gt.shape[1])) if len(cur_gt) == 0 else cur_gt
            if self._c('SAMPLE_ROI_BY_EACH_CLASS', False):
                max_overlaps, gt_assignment = self.get_max_iou_with_same_class(cur_roi, cur_labels, cur_gt[:, 0:7], cur_gt[:, -1].long())
            else:
                max_overlaps, gt_assignment = torch.max(iou3d_nms_utils.boxes_iou3d_gpu(cur_roi, cur_gt[:, 0:7]), dim=1)
            sel = self.subsample_rois(max_overlaps=max_overlaps)
            b_rois[i], b_labels[i], b_iou[i], b_scores[i] = cur_roi[sel], cur_labels[sel], max_overlaps[sel], cur_scores[sel]
            b_gt[i] = cur_gt[gt_assignment[sel]]
        return b_rois, b_gt, b_iou, b_scores, b_labels

    def subsample_rois(self, max_overlaps):
        n = self._c('ROI_PER_IMAGE')
        fg_per_image = int(np.round(self._c('FG_RATIO') * n))
        fg_thresh = min(self._c('REG_FG_THRESH'), self._c('CLS_FG_THRESH'))
        fg_inds = (max_overlaps >= fg_thresh).nonzero().view(-1)
        easy_bg = (max_overlaps < self._c('CLS_BG_THRESH_LO')).nonzero().view(-1)
        hard_bg = ((max_overlaps < self._c('REG_FG_THRESH')) & (max_overlaps >= self._c('CLS_BG_THRESH_LO'))).nonzero().view(-1)
        n_fg, n_bg = fg_inds.numel(), hard_bg.numel() + easy_bg.numel()
        if n_fg > 0 and n_bg > 0:
            fg_this = min(fg_per_image, n_fg)
            perm = torch.from_numpy(np.random.permutation(n_fg)).to(max_overlaps.device).long()
            fg_inds = fg_inds[perm[:fg_this]]
            bg_inds = self.sample_bg_inds(hard_bg, easy_bg, n - fg_this, self._c('HARD_BG_RATIO'))
        elif n_fg > 0 and n_bg == 0:
            r = torch.from_numpy(np.floor(np.random.rand(n) * n_fg)).to(max_overlaps.device).long()
            fg_inds = fg_inds[r]
            bg_inds = fg_inds[fg_inds < 0]
        elif n_bg > 0 and n_fg == 0:
            bg_inds = self.sample_bg_inds(hard_bg, easy_bg, n, self._c('HARD_BG_RATIO'))
        else:
            raise NotImplementedError('no RoIs to sample: FG=%d, BG=%d' % (n_fg, n_bg))
        return torch.cat((fg_inds, bg_inds), dim=0)

    @staticmethod
    def sample_bg_inds(hard_bg_inds, easy_bg_inds, bg_rois_per_this_image, hard_bg_ratio):
        dev = hard_bg_inds.device

        def draw(pool, k):
            return pool[torch.randint(low=0, high=pool.numel(), size=(k,)).long().to(dev)]

        if hard_bg_inds.numel() > 0 and easy_bg_inds.numel() > 0:
            n_hard = min(int(bg_rois_per_this_image * hard_bg_ratio), len(hard_bg_inds))
            hard = draw(hard_bg_inds, n_hard)
            easy = draw(easy_bg_inds, bg_rois_per_this_image - n_hard)
            return torch.cat([hard, easy], dim=0)
        if hard_bg_inds.numel() > 0:
            return draw(hard_bg_inds, bg_rois_per_this_image)
        if easy_bg_inds.numel() > 0:
            return draw(easy_bg_inds, bg_rois_per_this_image)
        raise NotImplementedError

    @staticmethod
    def get_max_iou_with_same_class(rois, roi_labels, gt_boxes, gt_labels):
        max_overlaps = rois.new_zeros(rois.shape[0])
        gt_assignment = roi_labels.new_zeros(roi_labels.shape[0])
        for k in range(gt_labels.min().item(), gt_labels.max().item() + 1):
            roi_mask, gt_mask = roi_labels == k, gt_labels == k
            if roi_mask.sum() > 0 and gt_mask.sum() > 0:
                orig = gt_mask.nonzero().view(-1)
                cur_max, cur_arg = torch.max(iou3d_nms_utils.boxes_iou3d_gpu(rois[roi_mask], gt_boxes[gt_mask]), dim=1)
                max_overlaps[roi_mask] = cur_max
                gt_assignment[roi_mask] = orig[cur_arg]
        return max_overlaps, gt_assignment
