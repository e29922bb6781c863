"""Debug aid: train-mode set abstraction, hand-written kernels vs the Conv2d path, per-parameter error report."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seevcn_amd  # noqa: F401
import seevcn_amd.synth as synth
from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as pm

cuda = torch.device("cuda:0")
c_in, mlps, nsamples = int(sys.argv[1]), [[64, 64], [32, 64]], [16, 32]
pts, _ = synth.make_scene_batch(3, seed=2000, n_az=60)
counts = np.bincount(pts[:, 0].astype(int), minlength=3)
xyz = np.ascontiguousarray(pts[:, 1:4])
rng = np.random.default_rng(c_in + 5)
qcnt = [700, 513, 64]
starts = np.cumsum(counts) - counts
new = np.concatenate([xyz[starts[b]:starts[b] + counts[b]][rng.integers(0, counts[b], q)] + rng.normal(0, 0.3, (q, 3)) for b, q in enumerate(qcnt)]).astype(np.float32)
new[7] = [500, 500, 500]
feats = rng.normal(size=(len(xyz), c_in)).astype(np.float32) if c_in else None
torch.manual_seed(c_in)
m1 = pm.StackSAModuleMSG(radii=[0.4, 1.2], nsamples=nsamples, mlps=[[c_in] + list(x) for x in mlps], use_xyz=True, pool_method='max_pool').to(cuda).train()
mode = os.environ.get("SA_DBG_BN", "")
with torch.no_grad():
    for mod in m1.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            if "w" in mode:
                mod.weight.uniform_(0.5, 1.5)
            if "n" in mode:
                mod.weight[::5] *= -1.0
            if "b" in mode:
                mod.bias.uniform_(-0.3, 0.3)
m2 = copy.deepcopy(m1)
t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(cuda) if dt is None else torch.tensor(a, dtype=dt, device=cuda)
args = (t(xyz), t(counts.tolist(), torch.int32), t(new), t(qcnt, torch.int32))
f1 = t(feats).requires_grad_(True) if c_in else None
f2 = t(feats).requires_grad_(True) if c_in else None
_, a = m1(*args, features=f1)
pm.TRAIN_SA_OFF = True
_, b = m2(*args, features=f2)
pm.TRAIN_SA_OFF = False
print("out maxdiff", float((a - b).abs().max()), float(b.abs().max()))
w = torch.from_numpy(rng.normal(size=tuple(a.shape)).astype(np.float32)).to(cuda)
sel = int(sys.argv[2]) if len(sys.argv) > 2 else -1
if sel >= 0:                      # gradient through one scale only
    w[:, :64] *= (sel == 0)
    w[:, 64:] *= (sel == 1)
(a * w).sum().backward()
(b * w).sum().backward()
for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
    x, y = p1.grad.reshape(p1.shape[0], -1).cpu().numpy(), p2.grad.reshape(p2.shape[0], -1).cpu().numpy()
    err = np.abs(x - y)
    print(n1, x.shape, "max|ref|", float(np.abs(y).max()), "max err", float(err.max()), "rel-to-max", float(err.max() / (np.abs(y).max() + 1e-30)))
    if x.ndim == 2 and x.shape[1] > 8:
        print("   per-column max err / col max:", np.round(err.max(0) / (np.abs(y).max(0) + 1e-30), 4)[:80])
if c_in:
    x, y = f1.grad.cpu().numpy(), f2.grad.cpu().numpy()
    err = np.abs(x - y)
    print("feature grad max|ref|", float(np.abs(y).max()), "max err", float(err.max()), "rows wrong", int((err.max(1) > 1e-3 * np.abs(y).max()).sum()), "of", len(y))
