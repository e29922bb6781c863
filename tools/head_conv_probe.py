#!/usr/bin/env python3
"""CenterHead's 36 branches (6 heads x center / center_z / dim / rot / vel / hm; conv3x3 64->64 + BN + ReLU, conv3x3 64->k) on one 180 x 180 map:
36 separate MIOpen convolutions per layer against ONE 64 -> 2304 convolution (layer 1) and ONE grouped convolution (layer 2, groups = 36, k padded to 3).
Forward + backward, events, median of 20."""
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
H = W = 180
x = torch.randn(1, 64, H, W, device=dev, requires_grad=True)
w1 = [torch.randn(64, 64, 3, 3, device=dev, requires_grad=True) for _ in range(36)]
w2 = [torch.randn(k, 64, 3, 3, device=dev, requires_grad=True) for _ in range(6) for k in (2, 1, 3, 2, 2, 2)]

def timed(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[n // 2]

def sep1():
    ys = [F.conv2d(x, w, padding=1) for w in w1]
    sum(y.sum() for y in ys).backward()
def merged1():
    y = F.conv2d(x, torch.cat(w1, 0), padding=1)
    y.sum().backward()
h = [torch.randn(1, 64, H, W, device=dev, requires_grad=True) for _ in range(36)]
def sep2():
    ys = [F.conv2d(hh, w, padding=1) for hh, w in zip(h, w2)]
    sum(y.sum() for y in ys).backward()
hcat = torch.randn(1, 2304, H, W, device=dev, requires_grad=True)
def grouped2():
    wp = torch.cat([F.pad(w, (0, 0, 0, 0, 0, 0, 0, 3 - w.shape[0])) for w in w2], 0)
    y = F.conv2d(hcat, wp, padding=1, groups=36)
    y.sum().backward()
def grouped2_cl():
    wp = torch.cat([F.pad(w, (0, 0, 0, 0, 0, 0, 0, 3 - w.shape[0])) for w in w2], 0)
    y = F.conv2d(hcat.contiguous(memory_format=torch.channels_last), wp.contiguous(memory_format=torch.channels_last), padding=1, groups=36)
    y.sum().backward()
def dense2():
    wd = torch.zeros(108, 2304, 3, 3, device=dev)
    y = F.conv2d(hcat, wd, padding=1)
    y.sum().backward()
bn_w = torch.ones(2304, device=dev, requires_grad=True); bn_b = torch.zeros(2304, device=dev, requires_grad=True)
rm, rv = torch.zeros(2304, device=dev), torch.ones(2304, device=dev)
def bn_merged():
    y = F.relu(F.batch_norm(hcat, rm, rv, bn_w, bn_b, True, 0.1, 1e-5))
    y.sum().backward()
def bn_sep():
    ys = [F.relu(F.batch_norm(hh, rm[:64], rv[:64], bn_w[:64], bn_b[:64], True, 0.1, 1e-5)) for hh in h]
    sum(y.sum() for y in ys).backward()
for name, fn in (("layer 1: 36 separate 64->64", sep1), ("layer 1: one 64->2304", merged1), ("layer 2: 36 separate 64->k", sep2), ("layer 2: grouped (36 x 64->3)", grouped2),
                 ("layer 2: grouped, channels_last", grouped2_cl), ("layer 2: dense 2304->108", dense2), ("BN+ReLU: 36 separate", bn_sep), ("BN+ReLU: one over 2304", bn_merged)):
    try:
        print(f"{name:36s} {timed(fn):8.3f} ms fwd+bwd", flush=True)
    except Exception as ex:
        print(f"{name:36s} failed: {type(ex).__name__}: {str(ex)[:100]}", flush=True)
