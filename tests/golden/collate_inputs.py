"""Three ragged samples in the per-frame dict layout DatasetTemplate.__getitem__ produces (dataset.py:120-172)."""
import numpy as np


def make_samples():
    rng = np.random.default_rng(21)
    samples = []
    for i, (p, v, g) in enumerate([(50, 7, 3), (31, 4, 0), (77, 9, 5)]):
        samples.append({
            'points': rng.normal(size=(p, 4)).astype(np.float32),
            'voxels': rng.normal(size=(v, 5, 4)).astype(np.float32),
            'voxel_coords': rng.integers(0, 40, (v, 3)).astype(np.int32),
            'voxel_num_points': rng.integers(1, 6, v).astype(np.int32),
            'gt_boxes': rng.normal(size=(g, 8)).astype(np.float32),
            'frame_id': np.array('%06d' % i),
            'use_lead_xyz': True,
            'image_shape': np.array([375, 1242], np.int32),
            'points_2d': rng.normal(size=(p // 3, 2)).astype(np.float32),
            'depth_maps': rng.normal(size=(10 + i, 12 - i)).astype(np.float32),
        })
    return samples
