#!/usr/bin/env python3
"""The index side of a bench batch ALONE (nothing else on the GPU): build_network_index of VoxelBackBone8x on 16 voxelised KITTI-shaped scenes --
20 builds; under `rocprofv3 --kernel-trace --stats` the per-kernel durations are the kernels' own (in the step they run beside the trained side's
resident conv launches and are stretched by them).  Prints wall time per build (host + GPU + the one read) and, for comparison, the layer-by-layer build."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import seevcn_amd.synth as synth
from seevcn_amd import spconv
from seevcn_amd.pcdet.models import backbones_3d
from seevcn_amd.pcdet.ops import voxel_ops
from seevcn_amd.spconv import conv as sconv


def main():
    dev = torch.device("cuda:0")
    bs = 16
    pts, _ = synth.make_scene_batch(bs, seed=2000, n_az=384)
    g = dict(point_cloud_range=[0, -40, -3, 70.4, 40, 1], voxel_size=[0.05, 0.05, 0.1], grid_size=[1408, 1600, 40])
    feats, coords, _ = voxel_ops.voxelize_dynamic(torch.from_numpy(pts).to(dev), g["point_cloud_range"], g["voxel_size"], g["grid_size"], bs)
    net = backbones_3d.__all__['VoxelBackBone8x']({}, 3, g['grid_size']).to(dev).train()
    for batched in (True, False):
        sconv.BATCH_INDEX = batched

        def build():
            sp = spconv.SparseConvTensor(feats, coords, net.sparse_shape, bs)
            spconv.prebuild_rulebooks(net, sp, with_backward=True)
            return sp
        for _ in range(3):
            build()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            build()
        torch.cuda.synchronize()
        print(f"{'build_network_index' if batched else 'layer by layer     '}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per build (host + GPU, GPU otherwise idle), "
              f"{coords.shape[0]} voxels")


if __name__ == "__main__":
    main()
