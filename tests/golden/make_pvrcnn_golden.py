"""Generate tests/golden/pvrcnn_heads.npz by running the REFERENCE's own VoxelSetAbstraction, PointHeadSimple and PVRCNNHead
(+ RoIHeadTemplate, ProposalTargetLayer) on CPU, pv_rcnn.yaml head configuration with reduced sizes.  The compiled CUDA ops
under those classes are served by the oracle (tests/golden/_refimport.py:_install_oracle_ops).

Run only in the build container (needs /root/reference):  python tests/golden/make_pvrcnn_golden.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refimport as R  # noqa: E402

R.import_pcdet()
from easydict import EasyDict  # noqa: E402
from pcdet.models.backbones_3d.pfe.voxel_set_abstraction import VoxelSetAbstraction  # noqa: E402
from pcdet.models.dense_heads.point_head_simple import PointHeadSimple  # noqa: E402
from pcdet.models.roi_heads.pvrcnn_head import PVRCNNHead  # noqa: E402
import seevcn_amd.synth as synth  # noqa: E402
from seevcn_amd.pcdet import model_cfgs as C  # noqa: E402
from pvrcnn_inputs import make_inputs, SMALL  # noqa: E402

torch.set_num_threads(8)
inp = make_inputs()
pfe_cfg, ph_cfg, rh_cfg = (EasyDict(c) for c in C.pvrcnn_cfg(**SMALL))
vsa = VoxelSetAbstraction(pfe_cfg, voxel_size=[0.05, 0.05, 0.1], point_cloud_range=np.array(C.KITTI_RANGE, np.float32), num_bev_features=32,
                          num_rawpoint_features=4)
ph = PointHeadSimple(num_class=1, input_channels=vsa.num_point_features_before_fusion, model_cfg=ph_cfg)
rh = PVRCNNHead(input_channels=vsa.num_point_features, model_cfg=rh_cfg, num_class=1)
for m, seed in ((vsa, 11), (ph, 12), (rh, 13)):
    m.load_state_dict(R.seeded_state_dict(m, seed=seed))


def batch():
    return {
        'batch_size': 2, 'points': torch.from_numpy(inp['points']), 'gt_boxes': torch.from_numpy(inp['gt_boxes']),
        'spatial_features': torch.from_numpy(inp['spatial_features']), 'spatial_features_stride': 8,
        'multi_scale_3d_features': {k: SimpleNamespace(indices=torch.from_numpy(inp[k + '_indices']), features=torch.from_numpy(inp[k + '_features']))
                                    for k in ('x_conv3', 'x_conv4')},
        'batch_cls_preds': torch.from_numpy(inp['batch_cls_preds']), 'batch_box_preds': torch.from_numpy(inp['batch_box_preds']),
        'cls_preds_normalized': False,
    }


out = {}
# ---- train mode
for m in (vsa, ph, rh):
    m.train()
np.random.seed(7)
torch.manual_seed(7)
bd = rh(ph(vsa(batch())))
point_loss, tb1 = ph.get_loss()
rcnn_loss, tb2 = rh.get_loss()
fr = rh.forward_ret_dict
out.update(point_coords=bd['point_coords'].numpy(), point_features=bd['point_features'].detach().numpy(),
           point_features_before_fusion=bd['point_features_before_fusion'].detach().numpy(),
           point_cls_scores=bd['point_cls_scores'].detach().numpy(), point_cls_labels=ph.forward_ret_dict['point_cls_labels'].numpy(),
           point_loss=np.float32(point_loss.item()), point_pos_num=np.float32(tb1['point_pos_num']),
           train_rois=fr['rois'].numpy(), gt_of_rois=fr['gt_of_rois'].numpy(), gt_iou_of_rois=fr['gt_iou_of_rois'].numpy(),
           reg_valid_mask=fr['reg_valid_mask'].numpy(), rcnn_cls_labels=fr['rcnn_cls_labels'].numpy(), rcnn_cls=fr['rcnn_cls'].detach().numpy(),
           rcnn_reg=fr['rcnn_reg'].detach().numpy(),
           **{k: np.float32(v) for k, v in tb2.items()})
print("train: point_loss", point_loss.item(), tb1, "rcnn", tb2)
# ---- eval mode
for m in (vsa, ph, rh):
    m.eval()
with torch.no_grad():
    bd = rh(ph(vsa(batch())))
out.update(eval_rois=bd['rois'].numpy(), eval_roi_labels=bd['roi_labels'].numpy(), eval_batch_cls_preds=bd['batch_cls_preds'].numpy(),
           eval_batch_box_preds=bd['batch_box_preds'].numpy(), eval_point_features=bd['point_features'].numpy())
np.savez_compressed(os.path.join(HERE, "pvrcnn_heads.npz"), **out)
print("eval rois", bd['rois'].shape, os.path.getsize(os.path.join(HERE, "pvrcnn_heads.npz")))
