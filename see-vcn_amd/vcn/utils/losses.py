import torch


def geodesic_distance(m1, m2):
    """Rotation angle between two batches of rotation matrices (reference utils/losses.py:7-22)."""
    m = torch.bmm(m1, m2.transpose(1, 2))
    cos = (m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2] - 1) / 2
    cos = torch.min(cos, torch.ones_like(cos))
    cos = torch.max(cos, -torch.ones_like(cos))
    return torch.acos(cos)
