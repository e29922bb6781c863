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
            ignore = (ious > self._c('CLS_BG_THRESH')) & (ious < self._c('CLS_FG_THRESH'))
            cls_labels = torch.where(ignore, -torch.ones_like(cls_labels), cls_labels)        # a select: no index list from the mask
        elif st == 'roi_iou':
            bg, fg = self._c('CLS_BG_THRESH'), self._c('CLS_FG_THRESH')
            fg_mask, bg_mask = ious > fg, ious < bg
            interval = (fg_mask == 0) & (bg_mask == 0)
            cls_labels = (fg_mask > 0).float()
            cls_labels = torch.where(interval, (ious - bg) / (fg - bg), cls_labels)
        elif st == 'raw_roi_iou':
            cls_labels = ious
        else:
            raise NotImplementedError
        return {'rois': rois, 'gt_of_rois': gt_of_rois, 'gt_iou_of_rois': ious, 'roi_scores': scores, 'roi_labels': labels,
                'reg_valid_mask': reg_valid_mask, 'rcnn_cls_labels': cls_labels}

    def sample_rois_for_rcnn(self, batch_dict):
        """Same selection as the reference (:82-128) with ONE device -> host read instead of ~30 per scene: the RoI x ground-truth overlaps of every
        scene are computed on the device against ALL rows of the padded gt block (rows past the last non-zero one are masked out instead of being
        cut off after reading their number), the (B, P) maximum overlaps are read once, the fg / hard-bg / easy-bg draws run on the host in the
        reference's order (np.random.permutation, then torch.randint for hard and easy background, scene after scene -- the same numbers from the
        same seeds), and the chosen indices go back in one copy."""
        batch_size = batch_dict['batch_size']
        rois, roi_scores, roi_labels, gt_boxes = batch_dict['rois'], batch_dict['roi_scores'], batch_dict['roi_labels'], batch_dict['gt_boxes']
        n, G = self._c('ROI_PER_IMAGE'), gt_boxes.shape[1]
        dev = rois.device
        # valid gt rows: up to the last row whose entries do not sum to zero (reference :95-98); an empty scene keeps one all-zero row
        pos = torch.arange(G, device=dev).view(1, -1)
        last = ((gt_boxes.sum(dim=2) != 0).long() * (pos + 1)).max(dim=1)[0]                  # (B,) rows kept
        valid = pos < torch.clamp(last, min=1).view(-1, 1)                                   # (B, G)
        by_class = self._c('SAMPLE_ROI_BY_EACH_CLASS', False)
        # all scenes in one launch (sv_boxes_iou3d_batch) and one masked maximum: the reference's loop over scenes ran ~35 launches per scene here
        iou = iou3d_nms_utils.boxes_iou3d_batch(rois, gt_boxes)                               # (B, P, G)
        ok = valid.unsqueeze(1)                                                              # (B, 1, G)
        if by_class:                                                                         # get_max_iou_with_same_class (:210-230) as a mask
            ok = ok & (roi_labels.unsqueeze(2) == gt_boxes[:, :, -1].long().unsqueeze(1))
        best, arg = torch.where(ok, iou, -torch.ones_like(iou)).max(dim=2)
        none = best < 0                                                                      # no ground truth (of its class): overlap 0, row 0
        max_overlaps = torch.where(none, torch.zeros_like(best), best)                       # (B, P)
        gt_assignment = torch.where(none, torch.zeros_like(arg), arg)
        host = max_overlaps.cpu().numpy()                                                    # the one read
        sel = torch.from_numpy(np.stack([self.subsample_rois_host(host[i]) for i in range(batch_size)])).to(dev)      # (B, n)
        b_rois = torch.gather(rois, 1, sel.unsqueeze(-1).expand(-1, -1, rois.shape[-1]))
        b_gt = torch.gather(gt_boxes, 1, torch.gather(gt_assignment, 1, sel).unsqueeze(-1).expand(-1, -1, gt_boxes.shape[-1]))
        return b_rois, b_gt, torch.gather(max_overlaps, 1, sel), torch.gather(roi_scores, 1, sel), torch.gather(roi_labels, 1, sel)

    def subsample_rois_host(self, max_overlaps):
        """subsample_rois (:130-171) on a host array (P,) float32 -> (ROI_PER_IMAGE,) int64; float32 comparisons and the order of the random
        draws as in the reference."""
        n = self._c('ROI_PER_IMAGE')
        fg_per_image = int(np.round(self._c('FG_RATIO') * n))
        fg_thresh = np.float32(min(self._c('REG_FG_THRESH'), self._c('CLS_FG_THRESH')))
        lo, reg = np.float32(self._c('CLS_BG_THRESH_LO')), np.float32(self._c('REG_FG_THRESH'))
        fg_inds = np.nonzero(max_overlaps >= fg_thresh)[0]
        easy_bg = np.nonzero(max_overlaps < lo)[0]
        hard_bg = np.nonzero((max_overlaps < reg) & (max_overlaps >= lo))[0]
        n_fg, n_bg = fg_inds.size, hard_bg.size + easy_bg.size
        if n_fg > 0 and n_bg > 0:
            fg_this = min(fg_per_image, n_fg)
            fg_inds = fg_inds[np.random.permutation(n_fg)[:fg_this]]
            bg_inds = self.sample_bg_inds_host(hard_bg, easy_bg, n - fg_this, self._c('HARD_BG_RATIO'))
        elif n_fg > 0 and n_bg == 0:
            fg_inds = fg_inds[np.floor(np.random.rand(n) * n_fg).astype(np.int64)]
            bg_inds = fg_inds[fg_inds < 0]
        elif n_bg > 0 and n_fg == 0:
            fg_inds = fg_inds[:0]
            bg_inds = self.sample_bg_inds_host(hard_bg, easy_bg, n, self._c('HARD_BG_RATIO'))
        else:
            raise NotImplementedError('no RoIs to sample: FG=%d, BG=%d' % (n_fg, n_bg))
        return np.concatenate((fg_inds, bg_inds)).astype(np.int64)

    @staticmethod
    def sample_bg_inds_host(hard_bg_inds, easy_bg_inds, bg_rois_per_this_image, hard_bg_ratio):
        def draw(pool, k):
            return pool[torch.randint(low=0, high=pool.size, size=(k,)).long().numpy()]

        if hard_bg_inds.size > 0 and easy_bg_inds.size > 0:
            n_hard = min(int(bg_rois_per_this_image * hard_bg_ratio), len(hard_bg_inds))
            return np.concatenate([draw(hard_bg_inds, n_hard), draw(easy_bg_inds, bg_rois_per_this_image - n_hard)])
        if hard_bg_inds.size > 0:
            return draw(hard_bg_inds, bg_rois_per_this_image)
        if easy_bg_inds.size > 0:
            return draw(easy_bg_inds, bg_rois_per_this_image)
        raise NotImplementedError
