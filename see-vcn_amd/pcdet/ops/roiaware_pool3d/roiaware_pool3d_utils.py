"""points_in_boxes_gpu with the reference's signature (detector3d/pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py:28-41)."""
import torch

from .... import _lib


def points_in_boxes_gpu(points, boxes):
    """points (B,M,3), boxes (B,T,7) -> (B,M) int32 box index of each point, background = -1"""
    assert boxes.shape[0] == points.shape[0]
    assert boxes.shape[2] == 7 and points.shape[2] == 3
    lib = _lib.load()
    _lib.require_cuda(points, boxes)
    batch_size, num_points, _ = points.shape
    p = points.contiguous().float()
    b = boxes.contiguous().float()
    out = torch.full((batch_size, num_points), -1, dtype=torch.int32, device=points.device)
    rc = lib.sv_points_in_boxes(_lib.ptr(b) if b.numel() else None, _lib.ptr(p) if p.numel() else None, batch_size, b.shape[1], num_points,
                                _lib.ptr(out) if out.numel() else None, _lib.stream())
    _lib.check(rc, "sv_points_in_boxes")
    return out
