import numpy as np
import torch

from ..... import _lib
from ....utils.common_utils import cfg_get


class AxisAlignedTargetAssigner(object):
    """Drop-in for the reference AxisAlignedTargetAssigner (target_assigner/axis_aligned_target_assigner.py:8-210) on the
    fused HIP kernel sv_assign_targets_axis_aligned: no (anchors x gt) IoU matrix, no python loop over batch x class.
    Built: the deterministic branch the SEE-VCN configs use (POS_FRACTION -1, MATCH_HEIGHT False, no multihead)."""

    def __init__(self, model_cfg, class_names, box_coder, match_height=False):
        super().__init__()
        anchor_generator_cfg = cfg_get(model_cfg, 'ANCHOR_GENERATOR_CONFIG')
        anchor_target_cfg = cfg_get(model_cfg, 'TARGET_ASSIGNER_CONFIG')
        self.box_coder = box_coder
        self.match_height = match_height
        self.class_names = np.array(class_names)
        self.anchor_class_names = [c['class_name'] for c in anchor_generator_cfg]
        pos_fraction = cfg_get(anchor_target_cfg, 'POS_FRACTION', -1.0)
        self.pos_fraction = pos_fraction if pos_fraction >= 0 else None
        self.sample_size = cfg_get(anchor_target_cfg, 'SAMPLE_SIZE', 512)
        self.norm_by_num_examples = cfg_get(anchor_target_cfg, 'NORM_BY_NUM_EXAMPLES', False)
        self.matched_thresholds = {c['class_name']: c['matched_threshold'] for c in anchor_generator_cfg}
        self.unmatched_thresholds = {c['class_name']: c['unmatched_threshold'] for c in anchor_generator_cfg}
        self.use_multihead = cfg_get(model_cfg, 'USE_MULTIHEAD', False)
        if self.pos_fraction is not None or self.match_height or self.use_multihead or self.norm_by_num_examples:
            raise NotImplementedError("only the deterministic nearest-BEV branch is built (POS_FRACTION<0, MATCH_HEIGHT False, "
                                      "NORM_BY_NUM_EXAMPLES False, no multihead)")
        assert box_coder.code_size == 7, "fused target assignment encodes 7-d residuals"
        self._dev = {}

    def _device_tables(self, all_anchors):
        dev = all_anchors[0].device
        key = (dev, tuple(a.data_ptr() for a in all_anchors))
        if key not in self._dev:
            per_set = [int(a.shape[3] * a.shape[4]) for a in all_anchors]
            offs = np.concatenate([[0], np.cumsum(per_set)]).astype(np.int32)
            set_class = np.array([int(np.nonzero(self.class_names == n)[0][0]) + 1 for n in self.anchor_class_names], np.int32)
            anchors = torch.cat([a.reshape(*a.shape[:3], -1, a.shape[-1]) for a in all_anchors], dim=-2).reshape(-1, 7).contiguous().float()
            self._dev[key] = dict(
                anchors=anchors, per_loc=int(offs[-1]),
                offs=torch.from_numpy(offs).to(dev), cls=torch.from_numpy(set_class).to(dev),
                mthr=torch.tensor([self.matched_thresholds[n] for n in self.anchor_class_names], dtype=torch.float32, device=dev),
                uthr=torch.tensor([self.unmatched_thresholds[n] for n in self.anchor_class_names], dtype=torch.float32, device=dev))
        return self._dev[key]

    def assign_targets(self, all_anchors, gt_boxes_with_classes):
        """all_anchors: [(nz,ny,nx,n_size,n_rot,7), ...] per class; gt_boxes (B, M, 8) -> dict of (B,A[,7]) tensors"""
        lib = _lib.load()
        _lib.require_cuda(gt_boxes_with_classes, *all_anchors)
        t = self._device_tables(all_anchors)
        gt = gt_boxes_with_classes.contiguous().float()
        B, G = gt.shape[0], gt.shape[1]
        A = t['anchors'].shape[0]
        dev = gt.device
        labels = torch.empty((B, A), dtype=torch.int32, device=dev)
        targets = torch.empty((B, A, 7), dtype=torch.float32, device=dev)
        weights = torch.empty((B, A), dtype=torch.float32, device=dev)
        scratch = torch.empty((max(B * G, 1),), dtype=torch.float32, device=dev)
        rc = lib.sv_assign_targets_axis_aligned(_lib.ptr(t['anchors']), A, t['per_loc'], len(all_anchors), _lib.ptr(t['offs']), _lib.ptr(t['cls']),
                                                _lib.ptr(t['mthr']), _lib.ptr(t['uthr']), _lib.ptr(gt) if G else None, B, G, _lib.ptr(scratch),
                                                _lib.ptr(labels), _lib.ptr(targets), _lib.ptr(weights), _lib.stream())
        _lib.check(rc, "sv_assign_targets_axis_aligned")
        return {'box_cls_labels': labels, 'box_reg_targets': targets, 'reg_weights': weights}
