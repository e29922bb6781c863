"""Generate tests/golden/vcn_train.npz: the REFERENCE's own VCN_VC (see/surface_completion/models/vcn/models/VCN_VC.py:110-214) and VCN_CN
(VCN_CN.py:110-156) modules in train mode (batch-statistics BatchNorm), forward + backward of a fixed scalar, evaluated at FLOAT64 -- the tight
pin for oracle/vcn_train.py (and, through it, for the device's training forward / backward).

The reference hard-codes `.float()` on its 3x3 rotation matrices (utils/transform.py:27-31,49-53) and a FloatTensor clamp (VCN_VC.py:15); to run
its modules at float64 this script maps torch.Tensor.float to torch.Tensor.double while they run (nothing of the reference is edited or copied).
Stored: inputs, outputs, running statistics after the step, and for every parameter gradient the entries oracle.vcn_train.sample_index names
(all entries for tensors up to 8 192 entries) plus the tensor's max |g|.

Run only in the build container (needs /root/reference):  python tests/golden/make_vcn_train_golden.py
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
from oracle.vcn_train import parity_loss, sample_index  # noqa: E402

torch.set_num_threads(8)
torch.Tensor.float = torch.Tensor.double
B = 8
clouds, boxes = synth.make_object_batch(B, seed=1000)
out = {"input": clouds, "gt_boxes": boxes}
for tag, cls, seed in (("vc", VCN_VC, 0), ("cn", VCN_CN, 1)):
    net = cls({})
    net.load_state_dict(R.seeded_state_dict(net, seed=seed))
    net = net.double().train()
    ret = net({"input": torch.from_numpy(clouds).double(), "gt_boxes": torch.from_numpy(boxes).double()})
    assert all(v.dtype == torch.float64 for v in ret.values())
    up = torch.randn(ret["coarse"].shape, generator=torch.Generator().manual_seed(1))
    parity_loss(ret, up).backward()
    for k, v in ret.items():
        out[f"{tag}.out.{k}"] = v.detach().numpy()
    for k, p in net.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.reshape(-1)
        out[f"{tag}.grad.{k}"] = g[sample_index(g.numel())].numpy()
        out[f"{tag}.gmax.{k}"] = np.float64(g.abs().max())
    for k, b in net.named_buffers():
        out[f"{tag}.buf.{k}"] = b.numpy()
    out[f"{tag}.seed"] = np.int64(seed)
np.savez_compressed(os.path.join(HERE, "vcn_train.npz"), **out)
print(len(out), "arrays;", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")
