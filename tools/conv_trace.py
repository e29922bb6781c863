#!/usr/bin/env python3
"""Per-wave timeline of one planned sparse-conv launch (sv_debug_conv_trace): where the launch's time goes -- prologue, main loop, epilogue,
idle tail -- and how evenly the SIMDs are loaded.  LAYER=subm3 (default) | subm4 | subm2 ...; prints a summary."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import seevcn_amd.synth as synth
from seevcn_amd import _lib
from seevcn_amd.pcdet.ops import voxel_ops
from seevcn_amd.spconv import functional as Fsp


def main():
    dev = torch.device("cuda:0")
    bs = 16
    pts, _ = synth.make_scene_batch(bs, seed=2000, n_az=384)
    g = dict(r=[0, -40, -3, 70.4, 40, 1], v=[0.05, 0.05, 0.1], grid=[1408, 1600, 40])
    feats, coords, _ = voxel_ops.voxelize_dynamic(torch.from_numpy(pts).to(dev), g["r"], g["v"], g["grid"], bs)
    shape = [41, 1600, 1408]
    want = os.environ.get("LAYER", "subm3")
    layers = [("subm1", 16, 16), ("spconv2", 16, 32), ("subm2", 32, 32), ("spconv3", 32, 64), ("subm3", 64, 64), ("spconv4", 64, 64), ("subm4", 64, 64)]
    c = coords
    for name, cin, cout in layers:
        if name.startswith("subm"):
            rb = Fsp.build_subm_rulebook(c, bs, shape, [3, 3, 3])
        else:
            rb = Fsp.build_sparse_rulebook(c, bs, shape, [3, 3, 3], [2, 2, 2], [0, 1, 1] if name == "spconv4" else [1, 1, 1])
        if name == want:
            break
        c, shape = rb.out_indices, rb.out_shape
    x = torch.randn(rb.n_in, cin, device=dev)
    w = torch.randn(rb.K, cin, cout, device=dev) * 0.1
    plan = rb.plan("fwd", cin, cout)
    ff = Fsp.fragment_cache.get(w)[0]
    for _ in range(3):
        Fsp.gather_gemm_planned(x, plan, ff, rb.n_out, rb.K, cin, cout)
    n_slots = 8 * 128 * 4 * 2 + 64
    buf = torch.zeros((n_slots, 8), dtype=torch.int64, device=dev)
    lib = _lib.load()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    Fsp.gather_gemm_planned(x, plan, ff, rb.n_out, rb.K, cin, cout)
    e.record()
    torch.cuda.synchronize()
    plain_us = s.elapsed_time(e) * 1e3
    lib.sv_debug_conv_trace(buf.data_ptr())
    s.record()
    Fsp.gather_gemm_planned(x, plan, ff, rb.n_out, rb.K, cin, cout)
    e.record()
    torch.cuda.synchronize()
    lib.sv_debug_conv_trace(None)
    t = buf.cpu().numpy()
    t = t[t[:, 3] != 0]
    if os.environ.get("DUMP"):                 # raw per-wave records for offline analysis: start, prologue cycles, loop cycles, end, HW_ID, XCC_ID, steps, (block << 8) | wave
        np.save(os.path.join(os.environ["DUMP"], f"conv_trace_{want}.npy"), t)
    xcc_id = (t[:, 5] & 0xf).astype(np.int64)
    hw = t[:, 4].astype(np.int64)
    # s_memtime counts shader cycles and is NOT comparable between CUs on this part (tools/wgrad_trace.py found offsets of millions of ticks inside one
    # XCD): every CU gets workgroups at the start of the launch, so its earliest wave start is the launch's start on that CU's clock
    cu_key = (xcc_id << 16) | (hw & 0xff00)
    _, cu_inv = np.unique(cu_key, return_inverse=True)
    base = np.full(cu_inv.max() + 1, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(base, cu_inv, t[:, 0])
    start, end = (t[:, 0] - base[cu_inv]).astype(float), (t[:, 3] - base[cu_inv]).astype(float)
    pro, loop = t[:, 1].astype(float), t[:, 2].astype(float)          # cycles in pass prologues / main loops, summed over the wave's passes
    life = end - start
    epi = life - pro - loop
    work = t[:, 6].astype(float)                                      # (tile, offset) steps
    span = float(end.max())
    nt, kq = min(cout, 64) // 16, cin // 16
    step_cycles = kq * 4 * nt * 32                                    # MFMA pipe cycles of one (tile, offset) step
    print(f"{want} {cin}->{cout} rows {rb.n_out} G {plan[2]} waves {len(t)}; plain launch {plain_us:.1f} us, traced launch {s.elapsed_time(e) * 1e3:.1f} us; span {span:.0f} cycles "
          f"(per-CU last end: median {np.median([end[cu_inv == i].max() for i in range(cu_inv.max() + 1)]):.0f})")
    print(f"wave life: mean {life.mean():.0f} median {np.median(life):.0f} p95 {np.percentile(life, 95):.0f} max {life.max():.0f};  of it prologues {pro.sum() / life.sum():.3f}, main loops "
          f"{loop.sum() / life.sum():.3f}, epilogues {epi.sum() / life.sum():.3f};  wave end p5 {np.percentile(end, 5):.0f} median {np.median(end):.0f} p95 {np.percentile(end, 95):.0f}")
    print(f"steps per wave: mean {work.mean():.1f} min {work.min():.0f} max {work.max():.0f};  loop cycles per step {loop.sum() / work.sum():.0f} (its MFMAs: {step_cycles} pipe cycles -> "
          f"{step_cycles * work.sum() / loop.sum():.3f} of the pipe per looping wave)")
    simd = (cu_key << 8) | (hw & 0x30)
    ids, inv = np.unique(simd, return_inverse=True)
    per_work = np.bincount(inv, weights=work)
    per_loop = np.bincount(inv, weights=loop)
    per_pro = np.bincount(inv, weights=pro)
    per_epi = np.bincount(inv, weights=epi)
    per_end = np.zeros(len(ids))
    np.maximum.at(per_end, inv, end)
    mfma = per_work * step_cycles
    print(f"SIMDs {len(ids)}; waves per SIMD: {np.bincount(np.bincount(inv))};  steps per SIMD max/mean {per_work.max() / per_work.mean():.3f};  SIMD last end p5 {np.percentile(per_end, 5):.0f} "
          f"median {np.median(per_end):.0f} max {per_end.max():.0f}")
    print(f"MFMA pipe cycles needed per SIMD: mean {mfma.mean():.0f} max {mfma.max():.0f} = {mfma.mean() / span:.3f} / {mfma.max() / span:.3f} of the span;  per SIMD, summed over its waves: "
          f"loops {per_loop.mean() / span:.2f} x span, prologues {per_pro.mean() / span:.2f} x, epilogues {per_epi.mean() / span:.2f} x")
    xw = np.bincount(xcc_id.astype(int), weights=work, minlength=8)[:8]
    print("steps per XCD:", (xw / xw.mean()).round(3))


if __name__ == "__main__":
    main()


