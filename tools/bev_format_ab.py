#!/usr/bin/env python3
"""A/B of the dense 2-D BEV backbone (MIOpen) in NCHW against channels_last: forward + backward of BaseBEVBackbone at the SECOND size
(16 x 256 x 200 x 176) and the PV-RCNN size (4 x 256 x 188 x 188).  Prints ms per forward + backward for both layouts."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import seevcn_amd  # noqa: F401
from seevcn_amd.pcdet import model_cfgs as C
from seevcn_amd.pcdet.models.backbones_2d import base_bev_backbone


def run(batch, h, w, channels_last, steps=8, warm=4):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = base_bev_backbone.BaseBEVBackbone(C.SECOND_BACKBONE_2D, 256).to(dev).train()
    x = torch.randn(batch, 256, h, w, device=dev)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    t = []
    for i in range(warm + steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = net({"spatial_features": x})["spatial_features_2d"]
        y.square().mean().backward()
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    return 1e3 * sorted(t[warm:])[len(t[warm:]) // 2], tuple(y.stride())


if __name__ == "__main__":
    for name, (b, h, w) in (("pvrcnn 4x256x188x188", (4, 188, 188)), ("second 16x256x200x176", (16, 200, 176))):
        for cl in (False, True):
            ms, st = run(b, h, w, cl)
            print(f"{name:24s} {'channels_last' if cl else 'NCHW':14s} {ms:8.2f} ms  out strides {st}", flush=True)
