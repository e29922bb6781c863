"""Timing of the VCN post-processing kernels on bench-shaped inputs (64 objects)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import seevcn_amd.synth as synth
from seevcn_amd.vcn.utils import sampling as S
from post_inputs import make_pairs
dev = torch.device("cuda:0")
partial, coarse = make_pairs(64, seed=5)
p, c = torch.from_numpy(partial).to(dev), torch.from_numpy(coarse).to(dev)
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for k in (10, 30):
    print(f"surface_select k={k}: {t(lambda: S.get_partial_mesh_batch_device(p, c, k=k)):.3f} ms")
surf, _ = S.get_partial_mesh_batch_device(p, c, k=30)
print(f"largest_cluster: {t(lambda: S.get_largest_cluster_batch_device(surf, eps=0.4, min_points=2)):.3f} ms")
