"""collate_batch / load_data_to_gpu (D24) against the reference's own DatasetTemplate.collate_batch (tests/golden/collate.npz)."""
import os

import numpy as np
import pytest
import torch

from collate_inputs import make_samples


def test_collate_batch_matches_reference_golden(golden_dir):
    from seevcn_amd.pcdet.datasets import collate_batch
    g = np.load(os.path.join(golden_dir, "collate.npz"))
    ret = collate_batch(make_samples())
    assert ret['batch_size'] == int(g['batch_size']) == 3
    keys = [k for k in g.files if k != 'batch_size']
    assert sorted(k for k, v in ret.items() if isinstance(v, np.ndarray)) == sorted(keys)
    for k in keys:
        assert ret[k].dtype == g[k].dtype and np.array_equal(ret[k], g[k]), k
    assert ret['points'].shape == (158, 5) and np.array_equal(np.unique(ret['points'][:, 0]), [0, 1, 2])
    assert ret['gt_boxes'].shape == (3, 5, 8) and not ret['gt_boxes'][1].any()          # the empty sample is all padding


@pytest.mark.gpu
def test_load_data_to_gpu_types(cuda):
    from seevcn_amd.pcdet.datasets import collate_batch, load_data_to_gpu
    bd = collate_batch(make_samples())
    load_data_to_gpu(bd)
    assert bd['points'].is_cuda and bd['points'].dtype == torch.float32 and bd['voxel_coords'].dtype == torch.float32
    assert bd['image_shape'].dtype == torch.int32 and isinstance(bd['frame_id'], np.ndarray) and bd['batch_size'] == 3
