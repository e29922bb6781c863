#!/usr/bin/env python3
"""cProfile of the host thread over 20 in-line bench steps (front + compute): where the ~6 ms of Python / ctypes / torch dispatch per step go."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        bench.run_step(model, opt, params, inputs, 1)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", "45")))
    st.sort_stats("cumtime").print_stats(int(os.environ.get("TOPC", "40")))


if __name__ == "__main__":
    main()
