"""The roiaware_pool3d helpers the hot path uses, with the reference's names and signatures (detector3d/pcdet/ops/roiaware_pool3d/
roiaware_pool3d_utils.py:9-41), over the pybind-level module roiaware_pool3d_cuda.  `points_in_boxes_cpu` keeps its name and
numpy / torch-CPU interface; the box test itself runs on the GPU (there is no CPU path in this build)."""
import numpy as np
import torch

from .... import _lib
from . import roiaware_pool3d_cuda


def points_in_boxes_cpu(points, boxes):
    """points (num_points,3), boxes (N,7) -> point_indices (N,num_points) int32 0/1 (box test with margin 1e-2, :9-25)."""
    assert boxes.shape[1] == 7
    assert points.shape[1] == 3
    is_numpy = isinstance(points, np.ndarray)
    p = (torch.from_numpy(points) if is_numpy else points).float().contiguous()
    b = (torch.from_numpy(boxes) if isinstance(boxes, np.ndarray) else boxes).float().contiguous()
    point_indices = p.new_zeros((b.shape[0], p.shape[0]), dtype=torch.int)
    roiaware_pool3d_cuda.points_in_boxes_cpu(b, p, point_indices)
    return point_indices.numpy() if is_numpy else point_indices


def points_in_boxes_gpu(points, boxes):
    """points (B,M,3), boxes (B,T,7) -> (B,M) int32 box index of each point, background = -1 (:28-41)"""
    assert boxes.shape[0] == points.shape[0]
    assert boxes.shape[2] == 7 and points.shape[2] == 3
    _lib.require_cuda(points, boxes)
    batch_size, num_points, _ = points.shape
    box_idxs_of_pts = torch.full((batch_size, num_points), -1, dtype=torch.int32, device=points.device)
    roiaware_pool3d_cuda.points_in_boxes_gpu(boxes.contiguous().float(), points.contiguous().float(), box_idxs_of_pts)
    return box_idxs_of_pts


class RoIAwarePool3d(torch.nn.Module):
    """Name kept so that `roiaware_pool3d_utils.RoIAwarePool3d` resolves (roiaware_pool3d_utils.py:44-53; built by partA2_head.py only,
    SURVEY.md 2: outside the hot path).  Constructing one fails at model-build time instead of at the first forward."""

    def __init__(self, out_size, max_pts_each_voxel=128):
        raise NotImplementedError("RoIAwarePool3d (PartA2 RoI-aware pooling) is outside the SEE-VCN hot path and not built; "
                                  "see INTEGRATION.md 'Unsupported surface'")
