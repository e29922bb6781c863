import torch


class AnchorGenerator(object):
    """Dense anchor grid per class, same constructor and output layout as the reference
    (dense_heads/target_assigner/anchor_generator.py:4-60): a list of (nz, ny, nx, n_size, n_rot, 7) tensors
    [x, y, z_centre, dx, dy, dz, rot] and the number of anchors per location of each set."""

    def __init__(self, anchor_range, anchor_generator_config):
        super().__init__()
        self.anchor_generator_cfg = anchor_generator_config
        self.anchor_range = anchor_range
        self.anchor_sizes = [c['anchor_sizes'] for c in anchor_generator_config]
        self.anchor_rotations = [c['anchor_rotations'] for c in anchor_generator_config]
        self.anchor_heights = [c['anchor_bottom_heights'] for c in anchor_generator_config]
        self.align_center = [c.get('align_center', False) for c in anchor_generator_config]
        assert len(self.anchor_sizes) == len(self.anchor_rotations) == len(self.anchor_heights)
        self.num_of_anchor_sets = len(self.anchor_sizes)

    def generate_anchors(self, grid_sizes, device=None):
        assert len(grid_sizes) == self.num_of_anchor_sets
        all_anchors, num_anchors_per_location = [], []
        r = self.anchor_range
        for grid_size, sizes, rotations, heights, align in zip(grid_sizes, self.anchor_sizes, self.anchor_rotations,
                                                               self.anchor_heights, self.align_center):
            num_anchors_per_location.append(len(rotations) * len(sizes) * len(heights))
            if align:
                x_stride, y_stride = (r[3] - r[0]) / grid_size[0], (r[4] - r[1]) / grid_size[1]
                x_offset, y_offset = x_stride / 2, y_stride / 2
            else:
                x_stride, y_stride = (r[3] - r[0]) / (grid_size[0] - 1), (r[4] - r[1]) / (grid_size[1] - 1)
                x_offset, y_offset = 0, 0
            # the same arange calls as the reference (:35-40) so that the float32 grid values are identical
            xs = torch.arange(r[0] + x_offset, r[3] + 1e-5, step=x_stride, dtype=torch.float32)
            ys = torch.arange(r[1] + y_offset, r[4] + 1e-5, step=y_stride, dtype=torch.float32)
            zs = torch.tensor(heights, dtype=torch.float32)
            size_t = torch.tensor(sizes, dtype=torch.float32).view(-1, 3)
            rot_t = torch.tensor(rotations, dtype=torch.float32)
            nz, ny, nx, ns, nr = len(zs), len(ys), len(xs), size_t.shape[0], len(rot_t)
            a = torch.empty((nz, ny, nx, ns, nr, 7), dtype=torch.float32)
            a[..., 0] = xs.view(1, 1, nx, 1, 1)
            a[..., 1] = ys.view(1, ny, 1, 1, 1)
            a[..., 2] = zs.view(nz, 1, 1, 1, 1)
            a[..., 3:6] = size_t.view(1, 1, 1, ns, 1, 3)
            a[..., 6] = rot_t.view(1, 1, 1, 1, nr)
            a[..., 2] += a[..., 5] / 2                      # bottom height -> box centre (:57)
            all_anchors.append(a.contiguous() if device is None else a.contiguous().to(device))
        return all_anchors, num_anchors_per_location
