#!/usr/bin/env python3
"""Every GEMM of one stage-A step (VCN_VC forward, 64 objects x 1024 points): shape, epilogue, time (events around each call, 20 repeats of the step) and
the fraction of the fp32 MFMA peak on the rows actually computed (distinct rows: read back once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs
from seevcn_amd.vcn.models import layers

dev = torch.device("cuda:0")
step, *_ = bench_configs.build("stageA", 0, dev)
for _ in range(5):
    step()
torch.cuda.synchronize()
calls, orig = [], layers.gemm
def traced(a, w, *args, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = orig(a, w, *args, **kw)
    e.record()
    m_dev = kw.get("m_dev")
    calls.append((a.shape[0], w.shape[0], w.shape[1], m_dev, kw.get("store", True), kw.get("group_max") is not None or (len(args) > 6 and args[6] is not None), s, e, kw.get("tag")))
    return out
layers.gemm = traced
import seevcn_amd.vcn.models.VCN_VC as VC
for mod in (VC,):
    if hasattr(mod, "gemm"):
        mod.gemm = traced
R = 20
for _ in range(R):
    step()
torch.cuda.synchronize()
per = len(calls) // R
print(f"{per} GEMM calls per step")
tot = 0.0
for i in range(per):
    M, N, K, m_dev, store, gmax, *_ , tag = calls[i]
    m_real = int(m_dev.item()) if m_dev is not None else M
    ms = sorted(calls[r * per + i][6].elapsed_time(calls[r * per + i][7]) for r in range(R))[R // 2]
    tf = 2.0 * m_real * N * K / ms / 1e9
    tot += ms
    print(f"  M {M:6d} (computed {m_real:6d}) N {N:5d} K {K:5d} {'store' if store else '     '} {'max' if gmax else '   '}  {ms * 1e3:7.1f} us  {tf:6.1f} TFLOP/s = {tf / 157.3:.2f} of peak   {tag or ''}")
print(f"sum {tot * 1e3:.1f} us")
