#!/usr/bin/env python3
"""Farthest point sampling at the PV-RCNN sizes (4 scenes x ~17k points -> 4096 keypoints; 2048 for KITTI): time per call and per round."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as pu


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    for counts, m in (([17000] * 4, 4096), ([17000] * 4, 2048), ([60000] * 2, 4096), ([17000] * 16, 2048), ([5000] * 4, 2048)):
        xyz = torch.from_numpy(rng.normal(size=(sum(counts), 3)).astype(np.float32) * 20).to(dev)
        cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
        for _ in range(2):
            pu.stack_farthest_point_sample(xyz, cnt, m)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            pu.stack_farthest_point_sample(xyz, cnt, m)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / 5
        print(f"scenes {len(counts)} x {counts[0]} pts -> {m}: {t:8.3f} ms  ({t / m * 1e3:.2f} us/round)  mode={os.environ.get('SEEVCN_FPS_MULTI', 'default')}")


if __name__ == "__main__":
    main()
