import torch

from .transform import rot_from_heading


def get_dims(pts):
    """(B,N,3) -> (B,3) axis-aligned extents (reference utils/bbox_utils.py:8-27)."""
    maxpts, _ = torch.max(pts, dim=1, keepdim=True)
    minpts, _ = torch.min(pts, dim=1, keepdim=True)
    return (maxpts - minpts).squeeze(1)


def get_bbox_from_keypoints(pts, gt_box):
    """(B,N,3), (B,7) -> (B,7) [centre of the bounds, extents in the GT-heading frame, GT heading] (bbox_utils.py:29-48)."""
    gt_rmat = rot_from_heading(gt_box[:, -1]).to(pts.device)
    maxpts, _ = torch.max(pts, dim=1, keepdim=True)
    minpts, _ = torch.min(pts, dim=1, keepdim=True)
    centre = (maxpts + minpts) / 2
    norm_pts = torch.bmm(pts - centre, gt_rmat.permute(0, 2, 1))
    return torch.cat([centre.squeeze(1), get_dims(norm_pts), gt_box[:, -1].unsqueeze(1).to(pts.device)], dim=1)
