#!/usr/bin/env python3
"""Farthest point sampling at the PV-RCNN sizes (4 scenes x ~17k points -> 4096 keypoints; 2048 for KITTI): time per call and per round, the bucket
kernel (csrc/fps_bucket.hip) beside the exhaustive ones, on an isotropic Gaussian cloud and on a lidar-like sweep."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as ps
import seevcn_amd.synth as synth


def cloud(kind, n, rng):
    if kind == "gauss":
        return (rng.normal(size=(n, 3)) * 20).astype(np.float32)
    pts, _ = synth.make_scene_batch(1, seed=int(rng.integers(1 << 30)), n_az=360)
    pts = pts[:, 1:4]
    pts = pts[rng.permutation(len(pts))]
    while len(pts) < n:
        pts = np.concatenate([pts, pts + rng.normal(size=pts.shape).astype(np.float32) * 0.02])
    return np.ascontiguousarray(pts[:n], dtype=np.float32)


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    for kind in ("sweep", "gauss"):
        for counts, m in (([17000] * 4, 4096), ([20000] * 4, 4096), ([17000] * 4, 2048), ([60000] * 2, 4096), ([17000] * 16, 2048), ([5000] * 4, 2048)):
            xyz = torch.from_numpy(np.concatenate([cloud(kind, c, rng) for c in counts])).to(dev)
            cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
            line = f"{kind:5s} scenes {len(counts):2d} x {counts[0]} pts -> {m}:"
            ref = None
            for bucketed in (True, False):
                for _ in range(2):
                    out = ps.stack_farthest_point_sampling(xyz, cnt, m, max(counts), bucketed)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5):
                    ps.stack_farthest_point_sampling(xyz, cnt, m, max(counts), bucketed)
                e.record()
                torch.cuda.synchronize()
                t = s.elapsed_time(e) / 5
                line += f"   {'buckets' if bucketed else 'exhaustive'} {t:7.3f} ms ({t / m * 1e3:.2f} us/round)"
                if ref is None:
                    ref = out
                else:
                    line += "   same picks" if torch.equal(ref, out) else "   PICKS DIFFER"
            print(line, flush=True)


if __name__ == "__main__":
    main()
