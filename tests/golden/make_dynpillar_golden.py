"""Generate tests/golden/dyn_pillar_vfe.npz by running the REFERENCE's own DynamicPillarVFE
(backbones_3d/vfe/dynamic_pillar_vfe.py:49-142) on CPU (torch_scatter served by the functional stub in _refimport),
cbgs_dyn_pp_centerpoint.yaml-style VFE config (USE_NORM, USE_ABSLOTE_XYZ, NUM_FILTERS [64, 64] -> two PFN layers), train mode.
Inputs are regenerated from the seed by the test; only outputs are stored.

Run only in the build container (needs /root/reference):  python tests/golden/make_dynpillar_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refimport as R  # noqa: E402

R.import_pcdet()
from easydict import EasyDict  # noqa: E402
from pcdet.models.backbones_3d.vfe.dynamic_pillar_vfe import DynamicPillarVFE  # noqa: E402
from dynpillar_inputs import CFG, GRID, RANGE, VOXEL, make_points  # noqa: E402

pts = make_points()
vfe = DynamicPillarVFE(model_cfg=EasyDict(CFG), num_point_features=4, voxel_size=VOXEL, grid_size=np.array(GRID), point_cloud_range=np.array(RANGE, np.float32))
vfe.load_state_dict(R.seeded_state_dict(vfe, seed=11))
vfe.train()
p = torch.from_numpy(pts).requires_grad_(False)
bd = vfe({'points': p, 'batch_size': 2})
feat = bd['pillar_features']
w = torch.from_numpy(np.random.default_rng(3).normal(size=tuple(feat.shape)).astype(np.float32))
(feat * w).sum().backward()
np.savez_compressed(os.path.join(HERE, "dyn_pillar_vfe.npz"), voxel_coords=bd['voxel_coords'].numpy(), pillar_features=feat.detach().numpy(),
                    grad_linear0=vfe.pfn_layers[0].linear.weight.grad.numpy(), grad_linear1=vfe.pfn_layers[1].linear.weight.grad.numpy(),
                    running_mean0=vfe.pfn_layers[0].norm.running_mean.numpy())
print(bd['voxel_coords'].shape, feat.shape, os.path.getsize(os.path.join(HERE, "dyn_pillar_vfe.npz")))
