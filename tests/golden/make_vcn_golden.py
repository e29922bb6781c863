"""Generate tests/golden/vcn_vc.npz and vcn_cn.npz by running the REFERENCE's own VCN_VC / VCN_CN
(see/surface_completion/models/vcn/models/VCN_VC.py:110-214, VCN_CN.py:111-156) on CPU.

Run only in the build container (needs /root/reference):  python tests/golden/make_vcn_golden.py
Weights are not stored: they are re-derived from the state-dict key names by seeding.seeded_state_dict
(seed 0), which the tests apply to the build's own modules.
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
from models.vcn.models.VCN_VC import VCN_VC  # noqa: E402
from models.vcn.models.VCN_CN import VCN_CN  # noqa: E402
import seevcn_amd.synth as synth  # noqa: E402

torch.set_num_threads(8)
clouds, boxes = synth.make_object_batch(4, seed=1000)

with torch.no_grad():
    net = VCN_VC({}).eval()
    net.load_state_dict(R.seeded_state_dict(net, seed=0))
    ret = net({"input": torch.from_numpy(clouds)})
    # intermediate: the two global features, for localising a mismatch
    np.savez_compressed(
        os.path.join(HERE, "vcn_vc.npz"), input=clouds,
        coarse=ret["coarse"].numpy(), reg_rot=ret["reg_rot"].numpy(), reg_centre=ret["reg_centre"].numpy())
    print("VCN_VC coarse", ret["coarse"].shape, float(ret["coarse"].abs().mean()))

    net = VCN_CN({}).eval()
    net.load_state_dict(R.seeded_state_dict(net, seed=0))
    ret = net({"input": torch.from_numpy(clouds), "gt_boxes": torch.from_numpy(boxes)})
    np.savez_compressed(os.path.join(HERE, "vcn_cn.npz"), input=clouds, gt_boxes=boxes,
                        coarse=ret["coarse"].numpy())
    print("VCN_CN coarse", ret["coarse"].shape, float(ret["coarse"].abs().mean()))
