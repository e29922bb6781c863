#!/usr/bin/env python3
"""Where the ONE host thread of the pipelined headline step spends its time: wall-clock stamps around the pieces of bench.run_step_prefetched
(forward + loss enqueue, input side up to its read, the rest of the trained side inside the read hook, the blocked part of the read, tables + plans,
stage A of the batch after next), medians over 30 steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from seevcn_amd import _lib


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    points, objects, scene, *_ = bench.make_inputs(0, dev)
    model = bench.build_model(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, fused=True)
    pre = bench.Prefetch(model, (points, objects, scene))
    T = {}

    def stamp(name, t0):
        T.setdefault(name, []).append(time.perf_counter() - t0)

    # the blocked part of the read: everything in host_ints after the hook
    orig_tolist = torch.Tensor.tolist
    orig_host_ints = _lib.host_ints

    def host_ints(tensors):
        hook = _lib.set_sync_hook(None)
        t0 = time.perf_counter()
        if hook is not None:
            hook()
        stamp("hook: backward + optimiser enqueue", t0)
        t1 = time.perf_counter()
        try:
            return orig_host_ints(tensors)
        finally:
            stamp("read: cat + copy + wait", t1)
            _lib.set_sync_hook(hook)

    _lib.host_ints = host_ints
    import seevcn_amd.spconv.functional as Fsp
    Fsp._lib.host_ints = host_ints
    orig_front_a, orig_front_b = model.front_a, model.front_b

    def front_a(*a):
        t0 = time.perf_counter()
        try:
            return orig_front_a(*a)
        finally:
            stamp("front_a: stage A enqueue", t0)

    def front_b(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig_front_b(*a, **k)
        finally:
            stamp("front_b total (incl. hook + read)", t0)

    model.front_a, model.front_b = front_a, front_b
    for _ in range(8):
        bench.run_step_prefetched(model, opt, params, pre, 1)
    torch.cuda.synchronize()
    T.clear()
    t_all = time.perf_counter()
    n = 30
    for _ in range(n):
        t0 = time.perf_counter()
        bench.run_step_prefetched(model, opt, params, pre, 1)
        stamp("step (host)", t0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) / n
    print(f"wall per step {wall * 1e3:.3f} ms")
    for k, v in T.items():
        print(f"{k:42s} {np.median(v) * 1e3:7.3f} ms  (x{len(v) / n:.0f} per step)")
    fb = np.median(T["front_b total (incl. hook + read)"]) - np.median(T["hook: backward + optimiser enqueue"]) - np.median(T["read: cat + copy + wait"])
    print(f"{'front_b own enqueue (vox + index, both phases)':42s} {fb * 1e3:7.3f} ms")
    rest = np.median(T["step (host)"]) - np.median(T["front_b total (incl. hook + read)"]) - np.median(T["front_a: stage A enqueue"])
    print(f"{'take() + forward + loss enqueue':42s} {rest * 1e3:7.3f} ms")


if __name__ == "__main__":
    main()
