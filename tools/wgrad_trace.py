#!/usr/bin/env python3
"""Per-wave timeline of one MFMA weight-gradient launch (sv_debug_wgrad_trace, the 64 -> 64 instance): how long a wave spends in its pass prologues
(table read + compaction + first operand loads) and in its MFMA loops, how long workgroups live, how many are resident, and how evenly the SIMDs are
loaded.  LAYER=subm3 (default) | subm4 | spconv4."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401 (LOAD=file analyses a dump without a GPU)

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
    assert cin == 64 and cout == 64, "the trace instance is the 64 -> 64 kernel"
    x = torch.randn(rb.n_in, cin, device=dev)
    dy = torch.randn(rb.n_out, cout, device=dev)
    K = rb.K
    wp = rb.wgrad_plan(cin, cout)              # None with SEEVCN_WGRAD_PLANNED=0: the chunked kernel
    for _ in range(3):
        Fsp.wgrad(x, rb.nbr_out, dy, K, cin, cout, plan=wp)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    Fsp.wgrad(x, rb.nbr_out, dy, K, cin, cout, plan=wp)
    e.record()
    torch.cuda.synchronize()
    plain_us = s.elapsed_time(e) * 1e3
    buf = torch.zeros((4 * 16384, 8), dtype=torch.int64, device=dev)
    lib = _lib.load()
    lib.sv_debug_wgrad_trace(buf.data_ptr())
    s.record()
    Fsp.wgrad(x, rb.nbr_out, dy, K, cin, cout, plan=wp)
    e.record()
    torch.cuda.synchronize()
    lib.sv_debug_wgrad_trace(None)
    t = buf.cpu().numpy()
    t = t[t[:, 3] != 0]
    if os.environ.get("DUMP"):
        np.save(os.environ["DUMP"], t)
    print(f"{want} {cin}->{cout} rows {rb.n_out}; plain launch {plain_us:.1f} us, traced launch (stage 1 + slab reduction) {s.elapsed_time(e) * 1e3:.1f} us")
    analyse(t, K)


def analyse(t, K=27):
    hw, xcc = (t[:, 4] & 0xffffffff).astype(np.int64), ((t[:, 4] >> 32) & 0xf).astype(np.int64)
    # s_memtime counts shader cycles and is NOT comparable between CUs on this part (offsets of millions of ticks inside one XCD): every CU gets
    # workgroups at the start of the launch, so its earliest wave start is the launch's start on that CU's clock
    cu = (xcc << 16) | (hw & 0xff00)            # gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    ids, inv = np.unique(cu, return_inverse=True)
    base = np.full(len(ids), np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(base, inv, t[:, 0])
    start, end, body = (t[:, 0] - base[inv]).astype(float), (t[:, 3] - base[inv]).astype(float), (t[:, 5] - base[inv]).astype(float)
    pro, loop = t[:, 1].astype(float), t[:, 2].astype(float)
    pairs, passes = (t[:, 6] & 0xffffffff).astype(float), (t[:, 6] >> 32).astype(float)
    k = (t[:, 7] >> 32).astype(int)
    span = float(end.max())
    life = end - start
    print(f"pairs {int(pairs.sum())} waves {len(t)} (workgroups {len(t) // 4}) on {len(ids)} CUs; span {span:.0f} ticks (last wave end; per-CU last end: median "
          f"{np.median([end[inv == i].max() for i in range(len(ids))]):.0f})")
    print(f"wave life: mean {life.mean():.0f} median {np.median(life):.0f} p95 {np.percentile(life, 95):.0f} max {life.max():.0f} ticks;  of it prologues {pro.sum() / life.sum():.3f}, "
          f"MFMA loops {loop.sum() / life.sum():.3f}, after the last pass (reduction through LDS behind barriers + slab store) {(end - body).sum() / life.sum():.3f}")
    print(f"per pass: prologue {pro.sum() / passes.sum():.0f} ticks, loop {loop.sum() / passes.sum():.0f} ticks, pairs {pairs.sum() / passes.sum():.1f};  loop ticks per 4-pair step "
          f"{loop.sum() / (pairs.sum() / 4):.0f} (16 MFMAs = 512 pipe cycles)")
    ev = np.concatenate([np.stack([start, np.ones_like(start)], 1), np.stack([end, -np.ones_like(end)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    alive = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0], append=ev[-1, 0])
    print(f"waves resident (time-weighted over the span): {float((alive * dt).sum() / span):.0f} of {4 * 4 * len(ids)} slots at four per SIMD;  second-round waves start at "
          f"{np.sort(start)[min(len(start) - 1, 4 * 4 * len(ids))]:.0f};  wave end p5 {np.percentile(end, 5):.0f} median {np.median(end):.0f} p95 {np.percentile(end, 95):.0f}")
    simd = (cu << 8) | (hw & 0x30)
    sids, sinv = np.unique(simd, return_inverse=True)
    per_pairs = np.bincount(sinv, weights=pairs)
    per_loop = np.bincount(sinv, weights=loop)
    per_pro = np.bincount(sinv, weights=pro)
    per_tail = np.bincount(sinv, weights=end - body)
    per_waves = np.bincount(sinv)
    per_end = np.zeros(len(sids))
    np.maximum.at(per_end, sinv, end)
    mfma = per_pairs / 4 * 16 * 32
    print(f"SIMDs {len(sids)}; waves per SIMD over the launch: min {per_waves.min()} mean {per_waves.mean():.1f} max {per_waves.max()};  pairs per SIMD max/mean {per_pairs.max() / per_pairs.mean():.3f};  "
          f"SIMD last end p5 {np.percentile(per_end, 5):.0f} median {np.median(per_end):.0f} max {per_end.max():.0f}")
    print(f"MFMA pipe cycles needed per SIMD: mean {mfma.mean():.0f} max {mfma.max():.0f} = {mfma.mean() / span:.3f} / {mfma.max() / span:.3f} of the span;  per SIMD, summed over its waves: "
          f"loop time {per_loop.mean() / span:.2f} x span, prologues {per_pro.mean() / span:.2f} x, after-last-pass {per_tail.mean() / span:.2f} x")
    # how busy is a SIMD's pipe while at least one of its waves is inside a loop?  (upper bound of what the loops could deliver)
    print(f"loop efficiency: MFMA cycles / loop ticks summed over waves = {mfma.sum() / loop.sum():.3f} (1.0 = a wave alone on the pipe never waits; 0.25 = four waves share it perfectly)")
    print("offset: pairs per workgroup | wave life median | loop ticks per step | prologue share | after-last-pass share")
    for kk in range(K):
        m = k == kk
        if m.any():
            print(f"  k={kk:2d}  {pairs[m].sum() / (m.sum() / 4):7.0f} | {np.median(life[m]):7.0f} | {loop[m].sum() / max(pairs[m].sum() / 4, 1):6.0f} | {pro[m].sum() / life[m].sum():.2f} | "
                  f"{(end - body)[m].sum() / life[m].sum():.2f}")
    xw = np.bincount(xcc, weights=pairs, minlength=8)[:8]
    print("pairs per XCD:", (xw / xw.mean()).round(3))


if __name__ == "__main__":
    if os.environ.get("LOAD"):
        analyse(np.load(os.environ["LOAD"]))
    else:
        main()
