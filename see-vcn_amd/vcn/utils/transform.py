"""Rigid-transform helpers with the reference's names (see/surface_completion/models/vcn/utils/transform.py:6-57), torch only."""
import torch


def rot_from_heading(heading):
    """(B) yaw -> (B,3,3) [[c,s,0],[-s,c,0],[0,0,1]] (anti-clockwise convention, transform.py:6-31)."""
    yaw = heading if isinstance(heading, torch.Tensor) else torch.as_tensor(heading)
    cosa, sina = torch.cos(yaw), torch.sin(yaw)
    zeros, ones = yaw.new_zeros(len(yaw)), yaw.new_ones(len(yaw))
    return torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()


def rotate_points_along_z(points, angle):
    """points (B,N,3+), angle (B): points[..., :3] @ rot_from_heading(angle) (transform.py:33-57)."""
    rot = rot_from_heading(angle)
    points_rot = torch.matmul(points[:, :, 0:3], rot)
    return torch.cat((points_rot, points[:, :, 3:]), dim=-1)


def vc_to_cn(points, gt_label):
    """view-centric -> canonical frame of the gt box: (points - centre) rotated by -heading (transform.py:91-113)."""
    assert gt_label.shape[1] == 7, f'gt_label wrong shape, should be (B 7) but given shape is {gt_label.shape}'
    assert points.shape[2] == 3, f'points wrong shape, should be (B N 3) but given shape is {points.shape}.'
    return rotate_points_along_z(points - gt_label[:, :3].unsqueeze(1), -gt_label[:, -1])


def cn_to_vc(points, gt_label):
    """canonical -> view-centric (transform.py:115-137)."""
    assert gt_label.shape[1] == 7, f'gt_label wrong shape, should be (B 7) but given shape is {gt_label.shape}'
    assert points.shape[2] == 3, f'points wrong shape, should be (B N 3) but given shape is {points.shape}.'
    return rotate_points_along_z(points, gt_label[:, -1]) + gt_label[:, :3].unsqueeze(1)


def normalize_scale(points, gt_label):
    """divide by the box length (transform.py:139-151)."""
    assert gt_label.shape[1] == 7 and points.shape[2] == 3
    return points / gt_label[:, 3].view(-1, 1, 1)


def restore_scale(points, gt_label):
    """multiply by the box length (transform.py:153-165)."""
    assert gt_label.shape[1] == 7 and points.shape[2] == 3
    return points * gt_label[:, 3].view(-1, 1, 1)
