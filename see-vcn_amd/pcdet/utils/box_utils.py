"""Box helpers with the reference's names (detector3d/pcdet/utils/box_utils.py)."""
import torch

from . import common_utils


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    """boxes (N,7+C): dims grow by extra_width (box_utils.py:182-195)."""
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    large = boxes3d.clone()
    large[:, 3:6] += common_utils.const_tensor(extra_width, boxes3d.device, boxes3d.dtype)[None, :]
    return large


def boxes_to_corners_3d(boxes3d):
    """(N,7) -> (N,8,3) corners in the reference's order (box_utils.py:28-52)."""
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    template = common_utils.const_tensor(([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                                          [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]), boxes3d.device, boxes3d.dtype) / 2
    corners3d = boxes3d[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    corners3d = common_utils.rotate_points_along_z(corners3d.view(-1, 8, 3), boxes3d[:, 6]).view(-1, 8, 3)
    corners3d += boxes3d[:, None, 0:3]
    return corners3d.numpy() if is_numpy else corners3d
