#!/usr/bin/env python3
"""Time of the one-launch conv plan builder (k_plan_region) on the submanifold tables of the bench workload; SEEVCN_PLAN_DEBUG drops passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import seevcn_amd.synth as synth
from seevcn_amd.pcdet.ops import voxel_ops
from seevcn_amd.spconv import functional as Fsp
from tools.spconv_micro import timeit


def main():
    dev = torch.device("cuda:0")
    pts, _ = synth.make_scene_batch(16, seed=2000, n_az=384)
    feats, coords, _ = voxel_ops.voxelize_dynamic(torch.from_numpy(pts).to(dev), [0, -40, -3, 70.4, 40, 1], [0.05, 0.05, 0.1], [1408, 1600, 40], 16)
    rb = Fsp.build_subm_rulebook(coords, 16, [41, 1600, 1408], [3, 3, 3])
    for g in (2, 4):
        t = timeit(lambda: Fsp.TablePlan(rb.nbr_out, rb.n_out, rb.K, rb.rows_out, rb.masks_out, g=g), 20)
        print(f"rows {rb.n_out} g {g}: {t:.1f} us  debug={os.environ.get('SEEVCN_PLAN_DEBUG', '0')}")


if __name__ == "__main__":
    main()
