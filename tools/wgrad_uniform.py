#!/usr/bin/env python3
"""Weight-gradient kernel on SYNTHETIC tables of uniform density (every offset has a pair for the same fraction of rows): separates the kernel's
own efficiency from the imbalance between the 27 offsets of a real rulebook (centre offset: every row; corners: one row in thirty)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from seevcn_amd.spconv import functional as Fsp


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    n, K, c = 139554, 27, 64
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, c, device=dev)
    dy = torch.randn(n, c, device=dev)
    for density in (1.0, 0.33, 0.1):
        keep = torch.rand((K, n), device=dev, generator=g) < density
        src = (torch.arange(n, device=dev, dtype=torch.int32)[None, :] + torch.arange(K, device=dev, dtype=torch.int32)[:, None] * 37) % n
        nbr = torch.where(keep, src, torch.full_like(src, -1)).contiguous()
        pairs = int(keep.sum())
        t = timeit(lambda: Fsp.wgrad(x, nbr, dy, K, c, c))
        print(f"density {density:4.2f}: pairs {pairs:8d}  wgrad {t:7.1f} us  {2.0 * pairs * c * c / t / 1e6:6.1f} TF")


if __name__ == "__main__":
    main()
