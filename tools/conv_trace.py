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
    lib.sv_debug_conv_trace(buf.data_ptr())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    Fsp.gather_gemm_planned(x, plan, ff, rb.n_out, rb.K, cin, cout)
    e.record()
    torch.cuda.synchronize()
    lib.sv_debug_conv_trace(None)
    t = buf.cpu().numpy()
    t = t[t[:, 3] != 0]
    xcc_id = (t[:, 5] & 0xf).astype(np.int64)
    # s_memtime counts shader cycles and is NOT comparable between CUs on this part (tools/wgrad_trace.py found offsets of millions of ticks inside one
    # XCD): every CU gets workgroups at the start of the launch, so its earliest wave start is the launch's start on that CU's clock
    cu_key = (xcc_id << 16) | (t[:, 4] & 0xff00)
    _, cu_inv = np.unique(cu_key, return_inverse=True)
    base = np.full(cu_inv.max() + 1, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(base, cu_inv, t[:, 0])
    start, pro, loop, end = (t[:, i] - base[cu_inv] for i in range(4))
    xcc_id = xcc_id.astype(int)
    total = end.max()
    print(f"{want} {cin}->{cout} rows {rb.n_out} G {plan[2]} waves {len(t)} launch {s.elapsed_time(e) * 1e3:.1f} us  span {total} ticks (s_memtime)")
    print(f"wave start: median {np.median(start):.0f} max {start.max():.0f} | prologue {np.median(pro - start):.0f} (p95 {np.percentile(pro - start, 95):.0f}) | "
          f"loop {np.median(loop - pro):.0f} (min {np.min(loop - pro):.0f} p95 {np.percentile(loop - pro, 95):.0f} max {np.max(loop - pro):.0f}) | "
          f"epilogue {np.median(end - loop):.0f} (p95 {np.percentile(end - loop, 95):.0f})")
    print(f"wave end: p5 {np.percentile(end, 5):.0f} median {np.median(end):.0f} p95 {np.percentile(end, 95):.0f} max {end.max():.0f}")
    work = t[:, 6].astype(float)
    print(f"work per wave (tile-offset steps): mean {work.mean():.1f} min {work.min():.0f} max {work.max():.0f}; loop ticks per step: median {np.median((loop - pro) / np.maximum(work, 1)):.0f}")
    # SIMD identity: xcc, se/sh/cu/simd from HW_ID (gfx9 layout: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13])
    hw, xcc = t[:, 4], t[:, 5] & 0xf
    simd = (xcc << 16) | (hw & 0xfff0 & ~0xc0)
    ids, inv = np.unique(simd, return_inverse=True)
    per_simd_work = np.bincount(inv, weights=work)
    per_simd_waves = np.bincount(inv)
    per_simd_end = np.zeros(len(ids))
    np.maximum.at(per_simd_end, inv, end)
    print(f"SIMDs used {len(ids)}; waves per SIMD: {np.bincount(per_simd_waves)}; work per SIMD mean {per_simd_work.mean():.0f} max {per_simd_work.max():.0f} "
          f"(max/mean {per_simd_work.max() / per_simd_work.mean():.3f}); SIMD end: p5 {np.percentile(per_simd_end, 5):.0f} median {np.median(per_simd_end):.0f} max {per_simd_end.max():.0f}")
    # pipe time needed: steps * 16 * NT MFMAs * 32 cycles  (s_memtime ticks at 100 MHz? print ratio instead)
    nt = min(cout, 64) // 16
    kq = cin // 16
    mfma_cycles = per_simd_work * kq * 4 * nt * 32
    print(f"MFMA cycles per SIMD: mean {mfma_cycles.mean():.0f} max {mfma_cycles.max():.0f}; ticks per MFMA cycle at the busiest SIMD {per_simd_end.max() / mfma_cycles.max():.4f}")
    if os.environ.get("MAP"):
        blk = (t[:, 7] >> 8).astype(int)
        wid = (t[:, 7] & 0xff).astype(int)
        simd_id, cu_id, sh_id, se_id = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
        sel = np.nonzero((blk % 8 == 0))[0]
        sel = sel[np.argsort(blk[sel] * 4 + wid[sel])]
        print("block/8 wid -> xcc se sh cu simd   (region 0)")
        for i in sel[:96]:
            print(f"  j={blk[i] // 8:3d} w={wid[i]} -> xcc {xcc[i]} se {se_id[i]} sh {sh_id[i]} cu {cu_id[i]:2d} simd {simd_id[i]}   hw=0x{hw[i]:x}")
        print("distinct (xcc) per region:", [sorted(set(xcc[blk % 8 == r].tolist())) for r in range(8)])
    xw = np.bincount(xcc.astype(int), weights=work, minlength=8)
    print("work per XCD:", (xw / xw.mean()).round(3))


if __name__ == "__main__":
    main()


def dump_mapping():
    """SEEVCN_TRACE_MAP=1: print how workgroups land on CUs / SIMDs (HW_ID fields) for region 0."""
    pass
