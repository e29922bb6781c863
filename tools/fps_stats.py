#!/usr/bin/env python3
"""Bucket statistics of csrc/fps_bucket.hip (library built with -DFB_STATS: tools/build_variant.sh fbstats "-DFB_STATS" fps_bucket; run with
SEEVCN_LIB=see-vcn_amd/lib/variants/libseevcn_hip_fbstats.so): per round (after the first 64), how many of a scene's 12 waves are touched, how many
buckets are re-computed, how often the tie branch runs."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from seevcn_amd import _lib
from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as ps
from fps_micro import cloud


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    lib = _lib.load()
    raw = ctypes.CDLL(lib._name) if hasattr(lib, "_name") else lib
    fn = raw.sv_fps_bucket_stats
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    out = (ctypes.c_ulonglong * 24)()
    print("# segment cycles come from s_memtime stamps, each waiting for every outstanding LDS / scalar-memory return of its wave: the instrumented round is\n"
          "# ~1.7x the production one (0.66 us = 1 580 cycles) and a segment that ends in an LDS write (publish) carries that wait -- a ranking, not a budget")
    for kind in ("sweep", "gauss"):
        for counts, m in (([17000] * 4, 4096), ([5000] * 4, 2048)):
            xyz = torch.from_numpy(np.concatenate([cloud(kind, c, rng) for c in counts])).to(dev)
            cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
            fn(out, 1)
            ps.stack_farthest_point_sampling(xyz, cnt, m, max(counts))
            fn(out, 1)
            rounds = out[0] / 12
            print(f"{kind} {len(counts)} x {counts[0]} -> {m}: per round and scene: waves touched {out[1] / rounds:.2f} of 12, buckets re-computed {out[2] / rounds:.2f}"
                  f" of {(counts[0] + 63) // 64}, tie branches {out[3] / rounds:.4f}, most buckets in one wave in any round {out[5]}")
            names = ["check", "update", "wave best", "publish", "barrier", "final", "bare stamp"]
            for t, label, cnt in ((0, "untouched", out[0] - out[1]), (1, "touched", out[1])):
                bare = out[8 + t * 7 + 6] / max(cnt, 1)
                print(f"    {label} wave-rounds ({cnt / out[0]:.2f} of all): cycles per segment minus a bare stamp ({bare:.0f}): "
                      + ", ".join(f"{names[k]} {out[8 + t * 7 + k] / max(cnt, 1) - bare:.0f}" for k in range(6))
                      + f"; sum {sum(out[8 + t * 7 + k] / max(cnt, 1) - bare for k in range(6)):.0f}")


if __name__ == "__main__":
    main()
