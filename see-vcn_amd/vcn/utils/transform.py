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
