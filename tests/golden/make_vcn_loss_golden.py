"""Generate tests/golden/vcn_loss.npz: the REFERENCE's own VCN_VC (see/surface_completion/models/vcn/models/VCN_VC.py:110-214) and
VCN_CN (VCN_CN.py:110-156) in train mode (batch-statistics BatchNorm) and its get_loss terms (:150-171).  The reference's CUDA-only ops are served by the
oracle: `chamfer.forward/backward` -> oracle/chamfer.py, `pointnet2_ops.furthest_point_sample / gather_operation` ->
oracle/pointnet2.py.  get_loss is called with training=False for dims / translation / rotation; the 'coarse' term is evaluated
with the reference's own misc.fps + ChamferDistanceL2 (the 'partial' term of the reference passes numpy arrays to the Chamfer
module, VCN_VC.py:172-174, and cannot run as written).

Run only in the build container (needs /root/reference):  python tests/golden/make_vcn_loss_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refimport as R  # noqa: E402

R.import_vcn()
from oracle import chamfer as och, pointnet2 as op2  # noqa: E402

ch = sys.modules["chamfer"]
ch.forward = lambda a, b: tuple(torch.from_numpy(x) for x in och.forward(a.detach().numpy(), b.detach().numpy()))
ch.backward = lambda a, b, i1, i2, g1, g2: tuple(torch.from_numpy(x) for x in och.backward(a.detach().numpy(), b.detach().numpy(), i1.numpy(), i2.numpy(),
                                                                                           g1.numpy(), g2.numpy()))
pu = sys.modules["pointnet2_ops.pointnet2_utils"]
pu.furthest_point_sample = lambda data, number: torch.from_numpy(np.stack([op2.farthest_point_sampling(d.numpy(), number) for d in data]).astype(np.int64))
pu.gather_operation = lambda feats, idx: torch.gather(feats, 2, idx.long().unsqueeze(1).expand(-1, feats.shape[1], -1))

from models.vcn.models.VCN_VC import VCN_VC  # noqa: E402
from models.vcn.utils import misc  # noqa: E402
from models.vcn.extensions.chamfer_dist import ChamferDistanceL2  # noqa: E402
from vcn_loss_inputs import make_batch  # noqa: E402

torch.set_num_threads(8)
inp, complete, gt = make_batch()
net = VCN_VC({})
net.load_state_dict(R.seeded_state_dict(net, seed=0))
net.train()
ret = net({"input": torch.from_numpy(inp)})
ld = net.get_loss(ret, {"gt_boxes": torch.from_numpy(gt), "training": False})
ds = misc.fps(torch.from_numpy(complete), ret["coarse"].shape[1])
coarse_loss = ChamferDistanceL2()(ret["coarse"], ds)
out = {k: ret[k].detach().numpy() for k in ("coarse", "reg_rot", "reg_centre")}
out.update({"loss_" + k: np.float32(float(v)) for k, v in ld.items()})
out["loss_coarse"] = np.float32(float(coarse_loss))

# VCN_CN (models/VCN_CN.py:110-156): train-mode forward with gt boxes and the 'coarse' loss term
from models.vcn.models.VCN_CN import VCN_CN  # noqa: E402
cn = VCN_CN({})
cn.load_state_dict(R.seeded_state_dict(cn, seed=1))
cn.train()
ret_cn = cn({"input": torch.from_numpy(inp), "gt_boxes": torch.from_numpy(gt)})
assert cn.get_loss(ret_cn, {"gt_boxes": torch.from_numpy(gt), "training": False}) == {}
ds_cn = misc.fps(torch.from_numpy(complete), ret_cn["coarse"].shape[1])
out["cn_coarse"] = ret_cn["coarse"].detach().numpy()
out["cn_loss_coarse"] = np.float32(float(ChamferDistanceL2()(ret_cn["coarse"], ds_cn)))
np.savez_compressed(os.path.join(HERE, "vcn_loss.npz"), **out)
print({k: (float(v) if v.ndim == 0 else v.shape) for k, v in out.items()})
