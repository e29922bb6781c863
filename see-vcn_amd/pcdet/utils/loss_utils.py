"""RPN losses with the reference's class names and call signatures (detector3d/pcdet/utils/loss_utils.py:9-206)."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib


FUSED_LOSS = os.environ.get("SEEVCN_FUSED_LOSS", "1") != "0"     # 0: the reference's chains of torch ops also on the GPU (A/B, tests)


class _FocalFunction(torch.autograd.Function):
    """SigmoidFocalClassificationLoss.forward and its derivative w.r.t. the logits as one launch each (sv_sigmoid_focal_loss); targets and
    weights are constants, as everywhere the reference calls it."""

    @staticmethod
    def forward(ctx, input, target, weights, alpha, gamma):
        lib = _lib.load()
        x, t = input.contiguous().float(), target.contiguous().float()
        w = None if weights is None else weights.contiguous().float()
        c = x.shape[-1]
        out = torch.empty_like(x)
        _lib.check(lib.sv_sigmoid_focal_loss(_lib.ptr(x), _lib.ptr(t), _lib.ptr(w), x.numel() // c, c, float(alpha), float(gamma), None, _lib.ptr(out),
                                             _lib.stream()), "sv_sigmoid_focal_loss")
        ctx.save_for_backward(x, t, w)
        ctx.alpha, ctx.gamma = float(alpha), float(gamma)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x, t, w = ctx.saved_tensors
        g = grad_out.contiguous().float()
        c = x.shape[-1]
        gin = torch.empty_like(x)
        _lib.check(lib.sv_sigmoid_focal_loss(_lib.ptr(x), _lib.ptr(t), _lib.ptr(w), x.numel() // c, c, ctx.alpha, ctx.gamma, _lib.ptr(g), _lib.ptr(gin),
                                             _lib.stream()), "sv_sigmoid_focal_loss (gradient)")
        return gin, None, None, None, None


class _SmoothL1Function(torch.autograd.Function):
    """WeightedSmoothL1Loss.forward and its derivative w.r.t. the prediction as one launch each (sv_weighted_smooth_l1_loss)."""

    @staticmethod
    def forward(ctx, input, target, code_weights, weights, beta):
        lib = _lib.load()
        x, t = input.contiguous().float(), target.contiguous().float()
        w = None if weights is None else weights.contiguous().float()
        c = x.shape[-1]
        out = torch.empty_like(x)
        _lib.check(lib.sv_weighted_smooth_l1_loss(_lib.ptr(x), _lib.ptr(t), _lib.ptr(code_weights), _lib.ptr(w), x.numel() // c, c, float(beta), None,
                                                  _lib.ptr(out), _lib.stream()), "sv_weighted_smooth_l1_loss")
        ctx.save_for_backward(x, t, code_weights, w)
        ctx.beta = float(beta)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x, t, cw, w = ctx.saved_tensors
        g = grad_out.contiguous().float()
        c = x.shape[-1]
        gin = torch.empty_like(x)
        _lib.check(lib.sv_weighted_smooth_l1_loss(_lib.ptr(x), _lib.ptr(t), _lib.ptr(cw), _lib.ptr(w), x.numel() // c, c, ctx.beta, _lib.ptr(g), _lib.ptr(gin),
                                                  _lib.stream()), "sv_weighted_smooth_l1_loss (gradient)")
        return gin, None, None, None, None


class SigmoidFocalClassificationLoss(nn.Module):
    """alpha-balanced sigmoid focal loss, un-reduced: (B, #anchors, #classes)  (loss_utils.py:9-72)"""

    def __init__(self, gamma: float = 2.0, alpha: float = 0.25):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    @staticmethod
    def sigmoid_cross_entropy_with_logits(input, target):
        # max(x, 0) - x * z + log(1 + exp(-|x|))
        return torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-torch.abs(input)))

    def forward(self, input, target, weights):
        if (FUSED_LOSS and input.is_cuda and input.dtype == torch.float32 and not target.requires_grad and not weights.requires_grad
                and target.shape == input.shape and weights.numel() * input.shape[-1] == input.numel()):
            return _FocalFunction.apply(input, target, weights, self.alpha, self.gamma)           # one launch per direction, same arithmetic
        p = torch.sigmoid(input)
        alpha_w = target * self.alpha + (1 - target) * (1 - self.alpha)
        pt = target * (1.0 - p) + (1.0 - target) * p
        loss = alpha_w * torch.pow(pt, self.gamma) * self.sigmoid_cross_entropy_with_logits(input, target)
        if weights.dim() == 2 or (weights.dim() == 1 and target.dim() == 2):
            weights = weights.unsqueeze(-1)
        assert weights.dim() == loss.dim()
        return loss * weights


class WeightedSmoothL1Loss(nn.Module):
    """code-wise weighted smooth-L1 (beta = 1/9), un-reduced (B, #anchors, #codes)  (loss_utils.py:75-136)"""

    def __init__(self, beta: float = 1.0 / 9.0, code_weights: list = None):
        super().__init__()
        self.beta = beta
        self.code_weights = None if code_weights is None else torch.from_numpy(np.array(code_weights, dtype=np.float32))

    @staticmethod
    def smooth_l1_loss(diff, beta):
        n = torch.abs(diff)
        return n if beta < 1e-5 else torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)

    def forward(self, input, target, weights=None):
        if (FUSED_LOSS and input.is_cuda and input.dtype == torch.float32 and not target.requires_grad and target.shape == input.shape
                and (weights is None or (not weights.requires_grad and weights.numel() * input.shape[-1] == input.numel()))
                and (self.code_weights is None or self.code_weights.numel() == input.shape[-1])):
            if self.code_weights is not None and self.code_weights.device != input.device:
                self.code_weights = self.code_weights.to(input.device)
            return _SmoothL1Function.apply(input, target, self.code_weights, weights, self.beta)
        target = torch.where(torch.isnan(target), input, target)  # ignore nan targets
        diff = input - target
        if self.code_weights is not None:
            if self.code_weights.device != diff.device:                  # moved once, not copied host -> device at every call
                self.code_weights = self.code_weights.to(diff.device)
            diff = diff * self.code_weights.view(1, 1, -1)
        loss = self.smooth_l1_loss(diff, self.beta)
        if weights is not None:
            assert weights.shape[0] == loss.shape[0] and weights.shape[1] == loss.shape[1]
            loss = loss * weights.unsqueeze(-1)
        return loss


class WeightedCrossEntropyLoss(nn.Module):
    """anchor-weighted softmax cross entropy on one-hot targets, un-reduced (B, #anchors)  (loss_utils.py:181-206)"""

    def forward(self, input, target, weights):
        return F.cross_entropy(input.permute(0, 2, 1), target.argmax(dim=-1), reduction='none') * weights


def get_corner_loss_lidar(pred_bbox3d, gt_bbox3d):
    """(N,7),(N,7) -> (N,) smooth-L1 (beta 1) corner distance, min over the gt box and its flipped heading (loss_utils.py:209-232)."""
    from . import box_utils
    assert pred_bbox3d.shape[0] == gt_bbox3d.shape[0]
    pred = box_utils.boxes_to_corners_3d(pred_bbox3d)
    gt = box_utils.boxes_to_corners_3d(gt_bbox3d)
    gt_flip = gt_bbox3d.clone()
    gt_flip[:, 6] += np.pi
    gtf = box_utils.boxes_to_corners_3d(gt_flip)
    dist = torch.min(torch.norm(pred - gt, dim=2), torch.norm(pred - gtf, dim=2))   # (N, 8)
    return WeightedSmoothL1Loss.smooth_l1_loss(dist, beta=1.0).mean(dim=1)


def neg_loss_cornernet(pred, gt, mask=None):
    """CornerNet focal loss on a Gaussian heat map (loss_utils.py:264-300)."""
    pos_inds = gt.eq(1).float()
    neg_inds = gt.lt(1).float()
    neg_weights = torch.pow(1 - gt, 4)
    pos_loss = torch.log(pred) * torch.pow(1 - pred, 2) * pos_inds
    neg_loss = torch.log(1 - pred) * torch.pow(pred, 2) * neg_weights * neg_inds
    if mask is not None:
        mask = mask[:, None, :, :].float()
        pos_loss, neg_loss = pos_loss * mask, neg_loss * mask
        num_pos = (pos_inds.float() * mask).sum()
    else:
        num_pos = pos_inds.float().sum()
    pos_loss, neg_loss = pos_loss.sum(), neg_loss.sum()
    # `-neg_loss if num_pos == 0 else -(pos_loss + neg_loss) / num_pos` of the reference as a select: the Python branch reads num_pos on the host.
    # With num_pos == 0 there is no positive term: pos_loss is 0 and both forms give -neg_loss
    return torch.where(num_pos == 0, -neg_loss, -(pos_loss + neg_loss) / torch.clamp(num_pos, min=1.0))


class FocalLossCenterNet(nn.Module):
    def forward(self, out, target, mask=None):
        return neg_loss_cornernet(out, target, mask=mask)


def _transpose_and_gather_feat(feat, ind):
    """(B,C,H,W), ind (B,K) -> (B,K,C) rows at flat positions ind (loss_utils.py:342-357)."""
    feat = feat.permute(0, 2, 3, 1).contiguous()
    feat = feat.view(feat.size(0), -1, feat.size(3))
    return feat.gather(1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), feat.size(2)))


class RegLossCenterNet(nn.Module):
    """masked L1 per code, normalised by the object count (loss_utils.py:303-385) -> (dim,)"""

    def forward(self, output, mask, ind=None, target=None):
        pred = output if ind is None else _transpose_and_gather_feat(output, ind)
        num = mask.float().sum()
        m = mask.unsqueeze(2).expand_as(target).float() * (~torch.isnan(target)).float()
        loss = torch.abs(pred * m - target * m).transpose(2, 0)
        loss = torch.sum(torch.sum(loss, dim=2), dim=1)
        return loss / torch.clamp_min(num, min=1.0)
