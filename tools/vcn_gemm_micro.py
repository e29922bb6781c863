#!/usr/bin/env python3
"""sv_gemm_bias_act at stage A's shapes (14 647 distinct rows of a 65 536-row capacity): the row count on the device (capacity-sized grid) against the
same rows known to the host; store / column-max epilogues; kernel time by events over 50 back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seevcn_amd.vcn.models import layers

dev = torch.device("cuda:0")
torch.manual_seed(0)
M_REAL, CAP, G = 14647, 65536, 64
row_group = torch.sort(torch.randint(0, G, (CAP,), device=dev, dtype=torch.int32)).values
m_dev = torch.tensor([M_REAL], dtype=torch.int32, device=dev)

def run(N, K, store, gmax, lazy, gb):
    M = CAP if lazy else M_REAL
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    gbias = torch.randn(G, N, device=dev) if gb else None
    gm = torch.full((G, N), float("-inf"), device=dev) if gmax else None
    rg = row_group[:M].contiguous()
    def call():
        return layers.gemm(a, w, b, layers.ACT_RELU if store else layers.ACT_NONE, group_bias=gbias, rows_per_group=1024, store=store, group_max=gm, row_group=rg,
                           m_dev=m_dev if lazy else None, tag="micro" if lazy else None)
    for _ in range(5):
        call()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        call()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 50 * 1e3
    return us, 2.0 * M_REAL * N * K / us / 1e6

for N, K in ((1024, 128), (1024, 512), (512, 256), (256, 128), (128, 64)):
    for store, gmax, gb in ((False, True, False), (True, False, False), (True, False, True), (True, True, False)):
        line = f"N {N:5d} K {K:4d} {'store' if store else '     '} {'max' if gmax else '   '} {'group bias' if gb else '          '}"
        for lazy in (True, False):
            us, tf = run(N, K, store, gmax, lazy, gb)
            line += f"   {'rows on device' if lazy else 'rows on host  '} {us:7.1f} us {tf:6.1f} TF ({tf / 157.3:.2f})"
        print(line, flush=True)
