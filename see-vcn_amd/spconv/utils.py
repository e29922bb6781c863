"""`spconv.utils` voxel generators as the reference binds them (detector3d/pcdet/datasets/processor/data_processor.py:15-60):

  spconv 1.x   VoxelGeneratorV2(voxel_size, point_cloud_range, max_num_points, max_voxels).generate(points) -> dict
               VoxelGenerator(...same...).generate(points) -> (voxels, coordinates, num_points_per_voxel)
  spconv 2.x   Point2VoxelCPU3d(vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels)
                   .point_to_voxel(tv.from_numpy(points)) -> three arrays with .numpy()

Same constructor keywords, same numpy in / numpy out contract (points (N,C) float32 -> voxels (V,max_points,C) zero padded,
coordinates (V,3) int32 [z,y,x], num_points_per_voxel (V) int32; first-come semantics in point order with both caps), but the
voxelisation itself is sv_voxelize_hard on the GPU (csrc/voxelize.hip): the reference runs spconv's CPU loop in the dataloader
workers.  The arrays go host -> device -> host here because that is this interface's contract; the training path hands device
tensors to pcdet.ops.voxel_ops.voxelize_hard directly (datasets/collate.py)."""
import numpy as np
import torch

from ..pcdet.ops import voxel_ops


class _HardVoxelizer:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels, device="cuda"):
        self._voxel_size = np.asarray(voxel_size, dtype=np.float32)
        self._point_cloud_range = np.asarray(point_cloud_range, dtype=np.float32)
        grid = (self._point_cloud_range[3:] - self._point_cloud_range[:3]) / self._voxel_size
        self._grid_size = np.round(grid).astype(np.int64)
        self._max_num_points, self._max_voxels, self._device = int(max_num_points), int(max_voxels), device

    voxel_size = property(lambda self: self._voxel_size)
    point_cloud_range = property(lambda self: self._point_cloud_range)
    grid_size = property(lambda self: self._grid_size)
    max_num_points_per_voxel = property(lambda self: self._max_num_points)

    def _run(self, points, max_voxels=None):
        pts = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float32)).to(self._device)
        assert pts.dim() == 2 and pts.shape[1] >= 3, "points must be (N, >=3)"
        mv = self._max_voxels if max_voxels is None else int(max_voxels)
        if pts.shape[0] == 0:
            c = pts.shape[1]
            return np.zeros((0, self._max_num_points, c), np.float32), np.zeros((0, 3), np.int32), np.zeros((0,), np.int32)
        voxels, coords, nump, nvox = voxel_ops.voxelize_hard(pts, 0, pts.shape[1], [pts.shape[0]], self._point_cloud_range, self._voxel_size,
                                                             self._grid_size, self._max_num_points, mv)
        n = int(nvox[0].item())
        return voxels[0, :n].cpu().numpy(), coords[0, :n].cpu().numpy(), nump[0, :n].cpu().numpy()


class VoxelGenerator(_HardVoxelizer):
    """spconv 1.0/1.1: generate() returns the 3-tuple (data_processor.py:50-51)."""

    def generate(self, points, max_voxels=None):
        return self._run(points, max_voxels)


class VoxelGeneratorV2(_HardVoxelizer):
    """spconv 1.2: generate() returns a dict (data_processor.py:46-49)."""

    def generate(self, points, max_voxels=None):
        voxels, coords, nump = self._run(points, max_voxels)
        return {'voxels': voxels, 'coordinates': coords, 'num_points_per_voxel': nump, 'voxel_num': len(voxels)}


class _ArrayView:
    """What point_to_voxel hands back: something with .numpy() (cumm.tensorview.Tensor in spconv 2.x, data_processor.py:56-59)."""

    def __init__(self, a):
        self._a = a

    def numpy(self):
        return self._a.copy()

    def numpy_view(self):
        return self._a


class Point2VoxelCPU3d(_HardVoxelizer):
    """spconv 2.x generator (data_processor.py:34-41,53-59); accepts a numpy array or anything with .numpy() (tv.from_numpy(points))."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels, device="cuda"):
        super().__init__(vsize_xyz, coors_range_xyz, max_num_points_per_voxel, max_num_voxels, device)
        self.num_point_features = int(num_point_features)

    def point_to_voxel(self, pc):
        points = pc if isinstance(pc, np.ndarray) else (pc.numpy() if hasattr(pc, "numpy") else np.asarray(pc))
        assert points.shape[1] == self.num_point_features, f"points have {points.shape[1]} features, generator built for {self.num_point_features}"
        return tuple(_ArrayView(a) for a in self._run(points))
