"""Same functions as the reference's detector3d/pcdet/ops/iou3d_nms/iou3d_nms_utils.py on libseevcn_hip.so."""
import torch

import os

from .... import _lib
from . import iou3d_nms_cuda

FUSED_IOU3D = os.environ.get("SEEVCN_FUSED_IOU3D", "1") != "0"     # 0: boxes_iou3d_gpu as the reference's chain of torch ops around the overlap kernel


def _pairs(boxes_a, boxes_b, iou):
    _lib.require_cuda(boxes_a, boxes_b)
    a = boxes_a[:, :7].contiguous().float()
    b = boxes_b[:, :7].contiguous().float()
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    (iou3d_nms_cuda.boxes_iou_bev_gpu if iou else iou3d_nms_cuda.boxes_overlap_bev_gpu)(a, b, out)
    return out


def boxes_iou_bev_cpu(boxes_a, boxes_b, device='cuda'):
    """(N,7),(M,7) numpy arrays or CPU tensors -> (N,M) rotated BEV IoU of the same type (reference iou3d_nms_utils.py:11-28, a host
    loop over iou3d_cpu.cpp there).  Keeps the name and the CPU-side interface; the pairs are evaluated by the same kernel as
    boxes_iou_bev -- this build has no CPU code path."""
    import numpy as np
    is_numpy = isinstance(boxes_a, np.ndarray)
    a = torch.from_numpy(boxes_a) if is_numpy else boxes_a
    b = torch.from_numpy(boxes_b) if isinstance(boxes_b, np.ndarray) else boxes_b
    assert not (a.is_cuda or b.is_cuda), 'Only support CPU tensors'
    assert a.shape[1] == 7 and b.shape[1] == 7
    out = _pairs(a.float().to(device), b.float().to(device), True).cpu()
    return out.numpy() if is_numpy else out


def boxes_iou_bev(boxes_a, boxes_b):
    """(N,7),(M,7) -> (N,M) rotated BEV IoU (reference iou3d_nms_utils.py:33-45)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return _pairs(boxes_a, boxes_b, True)


def boxes_overlap_bev(boxes_a, boxes_b):
    return _pairs(boxes_a, boxes_b, False)


def boxes_iou3d_batch(boxes_a, boxes_b):
    """(B,N,>=7), (B,M,>=7) -> (B,N,M) 3-D IoU of every scene's pairs in one launch (sv_boxes_iou3d_batch; seevcn extension: the reference calls
    boxes_iou3d_gpu scene by scene, ~22 launches each).  Same arithmetic as boxes_iou3d_gpu, operation by operation."""
    lib = _lib.load()
    _lib.require_cuda(boxes_a, boxes_b)
    assert boxes_a.dim() == 3 and boxes_b.dim() == 3 and boxes_a.shape[0] == boxes_b.shape[0] and boxes_a.shape[2] >= 7 and boxes_b.shape[2] >= 7
    a, b = boxes_a.contiguous().float(), boxes_b.contiguous().float()
    out = torch.empty((a.shape[0], a.shape[1], b.shape[1]), dtype=torch.float32, device=a.device)
    _lib.check(lib.sv_boxes_iou3d_batch(_lib.ptr(a), a.shape[1], a.shape[2], _lib.ptr(b), b.shape[1], b.shape[2], a.shape[0], _lib.ptr(out), _lib.stream()),
               "sv_boxes_iou3d_batch")
    return out


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7),(M,7) -> (N,M) 3-D IoU = BEV overlap x height overlap / union volume (reference :48-81)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    if FUSED_IOU3D:
        return boxes_iou3d_batch(boxes_a[None], boxes_b[None])[0]
    boxes_a_height_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    boxes_a_height_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    boxes_b_height_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    boxes_b_height_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = _pairs(boxes_a, boxes_b, False)
    max_of_min = torch.max(boxes_a_height_min, boxes_b_height_min)
    min_of_max = torch.min(boxes_a_height_max, boxes_b_height_max)
    overlaps_h = torch.clamp(min_of_max - max_of_min, min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def _nms(boxes, scores, thresh, pre_maxsize, normal, max_keep=None, padded=None, presorted=False):
    lib = _lib.load()
    _lib.require_cuda(boxes, scores)
    assert boxes.shape[1] == 7
    if presorted:                                          # the caller's boxes come out of a sorted top-k: no second sort (8-10 launches)
        order = None
        sorted_boxes = (boxes if pre_maxsize is None else boxes[:pre_maxsize]).contiguous().float()
    else:
        order = scores.sort(0, descending=True)[1]
        if pre_maxsize is not None:
            order = order[:pre_maxsize]
        sorted_boxes = boxes[order].contiguous().float()
    n = sorted_boxes.shape[0]
    dev = boxes.device
    keep = torch.empty((max(n, 1, padded or 0),), dtype=torch.int64, device=dev)
    num_out = torch.zeros((1,), dtype=torch.int32, device=dev)
    scratch = _lib.workspace.scratch("nms", lib.sv_nms_scratch_bytes(n), dev)
    rc = lib.sv_nms_prefix(_lib.ptr(sorted_boxes) if n else None, n, float(thresh), int(normal), n if max_keep is None else min(n, int(max_keep)),
                           _lib.ptr(scratch), _lib.ptr(keep), _lib.ptr(num_out), _lib.stream())
    _lib.check(rc, "sv_nms_prefix")
    if padded is not None:
        # no device -> host read: `padded` slots, the survivors first (score order), the rest index 0 with valid = False
        valid = torch.arange(padded, device=dev) < torch.clamp(num_out, max=padded)
        if n == 0:
            return torch.zeros((padded,), dtype=torch.int64, device=dev), valid
        kept = torch.where(valid, keep[:padded], torch.zeros_like(keep[:padded]))
        return (kept if order is None else order[kept]), valid
    kept = keep[:int(num_out.item())]
    return (kept if order is None else order[kept]).contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, max_keep=None, **kwargs):
    """Rotated NMS; returns (kept indices into `boxes` in score order, None) like the reference (:84-99).  max_keep (seevcn extension): stop
    after that many survivors -- the same prefix, for callers that truncate to NMS_POST_MAXSIZE anyway."""
    return _nms(boxes, scores, thresh, pre_maxsize, False, max_keep)


def nms_normal_gpu(boxes, scores, thresh, max_keep=None, **kwargs):
    return _nms(boxes, scores, thresh, None, True, max_keep)


def nms_gpu_padded(boxes, scores, thresh, slots, pre_maxsize=None, normal=False, presorted=False):
    """(seevcn extension) the first `slots` survivors of nms_gpu / nms_normal_gpu as a fixed-size result: (indices (slots,) int64, valid (slots,)
    bool) -- no device -> host read of the survivor count, for callers that fill a zero-padded (slots, .) block anyway
    (RoIHeadTemplate.proposal_layer, roi_head_template.py:46-102).  presorted: `boxes` are already in descending score order (a sorted top-k)."""
    return _nms(boxes, scores, thresh, pre_maxsize, normal, slots, padded=int(slots), presorted=presorted)
