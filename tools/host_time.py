#!/usr/bin/env python3
"""Host-side cost of one bench step: wall time the Python thread spends ENQUEUEING the trained side (compute) and the input side (front,
including its device -> host reads), against the GPU time of the same pieces.  Prints medians over 15 steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    points, objects, scene, *_ = bench.make_inputs(0, dev)
    inputs = (points, objects, scene)
    model = bench.build_model(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, fused=True)
    for _ in range(5):
        bench.run_step(model, opt, params, inputs, 1)
    torch.cuda.synchronize()
    rec = {"front_host": [], "front_gpu": [], "compute_host": [], "compute_gpu": []}
    for _ in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bd = model.front(*inputs, bench.SCENES_PER_GPU)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = bench.loss_fn(model.compute(bd))
        loss.backward()
        opt.step()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        rec["front_host"].append(t1 - t0)
        rec["front_gpu"].append(t2 - t0)
        rec["compute_host"].append(t3 - t2)
        rec["compute_gpu"].append(t4 - t2)
    for k, v in rec.items():
        print(f"{k:14s} {np.median(v) * 1e3:7.3f} ms")


if __name__ == "__main__":
    main()
