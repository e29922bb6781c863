"""Generate tests/golden/pillar_vfe.npz by running the REFERENCE's own PillarVFE (backbones_3d/vfe/pillar_vfe.py:52-123) and
PointPillarScatter (backbones_2d/map_to_bev/pointpillar_scatter.py:5-37) on CPU, pointpillar.yaml configuration, eval mode.
The hard voxels fed to them come from oracle/hard_voxelize.py (spconv's voxeliser is not installed).

Run only in the build container (needs /root/reference):  python tests/golden/make_pillar_golden.py
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
from pcdet.models.backbones_2d.map_to_bev.pointpillar_scatter import PointPillarScatter  # noqa: E402
from pcdet.models.backbones_3d.vfe.pillar_vfe import PillarVFE  # noqa: E402
import seevcn_amd.synth as synth  # noqa: E402
from oracle import hard_voxelize as ohv  # noqa: E402
from seevcn_amd.pcdet import model_cfgs as C  # noqa: E402

vs, rng_ = C.PP_VOXEL['VOXEL_SIZE'], C.PP_RANGE
vox, crd, nmp = [], [], []
for b in range(2):
    pts, _ = synth.make_scene(2000 + b, n_az=80)
    inten = np.random.default_rng(b).uniform(size=(len(pts), 1)).astype(np.float32)
    v, c, n = ohv.points_to_voxel(np.concatenate([pts, inten], 1), vs, rng_, 32, 1500)
    vox.append(v); nmp.append(n)
    crd.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
voxels, coords, nump = np.concatenate(vox), np.concatenate(crd), np.concatenate(nmp)
vfe = PillarVFE(model_cfg=EasyDict(C.PP_VFE), num_point_features=4, voxel_size=vs, point_cloud_range=np.array(rng_, np.float32)).eval()
vfe.load_state_dict(R.seeded_state_dict(vfe, seed=5))
grid = np.round((np.array(rng_[3:]) - np.array(rng_[:3])) / np.array(vs)).astype(np.int64)
sc = PointPillarScatter(model_cfg=EasyDict(C.PP_MAP_TO_BEV), grid_size=grid)
with torch.no_grad():
    bd = vfe({'voxels': torch.from_numpy(voxels), 'voxel_num_points': torch.from_numpy(nump), 'voxel_coords': torch.from_numpy(coords)})
    bd = sc(bd)
sf = bd['spatial_features'].numpy()
nzb, nzc, nzy, nzx = np.nonzero(sf)
np.savez_compressed(os.path.join(HERE, "pillar_vfe.npz"), voxels=voxels, voxel_coords=coords, voxel_num_points=nump,
                    pillar_features=bd['pillar_features'].numpy(), spatial_shape=np.array(sf.shape), spatial_sum=np.float64(sf.astype(np.float64).sum()),
                    spatial_nnz=np.int64(len(nzb)))
print(voxels.shape, bd['pillar_features'].shape, sf.shape, os.path.getsize(os.path.join(HERE, "pillar_vfe.npz")))
