#!/usr/bin/env python3
"""Micro-benchmark of the sparse-conv layers of VoxelBackBone8x on the bench workload (16 KITTI-shaped scenes).
Prints per-layer forward / backward-data / weight-grad times (HIP events) with pairs and achieved TFLOP/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import seevcn_amd.synth as synth
from seevcn_amd.pcdet.ops import voxel_ops
from seevcn_amd.spconv import functional as Fsp


_FLUSH = None


def timeit(fn, reps=10):
    """HIP-event time of fn in us. With COLD=1 a 1 GiB fill runs before every repetition (evicts L2 + Infinity Cache,
    like the dense BEV tensors do between two sparse layers of a real step) and only fn is bracketed by the events."""
    global _FLUSH
    cold = os.environ.get("COLD") == "1"
    for _ in range(3):
        fn()
    if not cold:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps * 1e3  # us
    if _FLUSH is None:
        _FLUSH = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
    tot = 0.0
    for i in range(reps):
        _FLUSH.fill_(float(i))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    bs = int(os.environ.get("BS", "16"))
    pts, _ = synth.make_scene_batch(bs, seed=2000, n_az=int(os.environ.get("N_AZ", "384")))
    if os.environ.get("SAME_SCENE"):          # measurement: every scene of the batch the same one -- the eight regions of a plan then carry identical work
        import numpy as np
        one = pts[pts[:, 0] == int(os.environ["SAME_SCENE"])]
        pts = np.concatenate([np.concatenate([np.full((len(one), 1), i, np.float32), one[:, 1:]], axis=1) for i in range(bs)], axis=0)
    g = dict(point_cloud_range=[0, -40, -3, 70.4, 40, 1], voxel_size=[0.05, 0.05, 0.1], grid_size=[1408, 1600, 40])
    p = torch.from_numpy(pts).to(dev)
    feats, coords, _ = voxel_ops.voxelize_dynamic(p, g["point_cloud_range"], g["voxel_size"], g["grid_size"], bs)
    print(f"voxelize: {timeit(lambda: voxel_ops.voxelize_dynamic(p, g['point_cloud_range'], g['voxel_size'], g['grid_size'], bs)):.1f} us  V={len(coords)}")
    shape = [41, 1600, 1408]
    layers = [("subm1", 16, 16), ("spconv2", 16, 32), ("subm2", 32, 32), ("spconv3", 32, 64), ("subm3", 64, 64),
              ("spconv4", 64, 64), ("subm4", 64, 64), ("down", 64, 128)]
    c = coords
    tot = 0.0
    only = os.environ.get("LAYER")
    mode = os.environ.get("MODE", "all")
    for name, cin, cout in layers:
        if name.startswith("subm"):
            tb = timeit(lambda: Fsp.build_subm_rulebook(c, bs, shape, [3, 3, 3]), 5)
            rb = Fsp.build_subm_rulebook(c, bs, shape, [3, 3, 3])
        else:
            ks, st, pd = ([3, 1, 1], [2, 1, 1], [0, 0, 0]) if name == "down" else ([3, 3, 3], [2, 2, 2], [0, 1, 1] if name == "spconv4" else [1, 1, 1])
            tb = timeit(lambda: Fsp.build_sparse_rulebook(c, bs, shape, ks, st, pd), 5)
            rb = Fsp.build_sparse_rulebook(c, bs, shape, ks, st, pd)
        pairs = int(rb.pair_counts().sum().item())
        K = rb.K
        x = torch.randn(rb.n_in, cin, device=dev)
        w = torch.randn(K, cin, cout, device=dev) * 0.1
        wt = w.permute(0, 2, 1).contiguous()
        dy = torch.randn(rb.n_out, cout, device=dev)
        if only and name != only:
            c, shape = rb.out_indices, rb.out_shape
            continue
        tf = td = tw = float("nan")
        # FIN=1: forward and weight gradient read x through an input transform (sv_conv_next_input_norm: BatchNorm + ReLU of the layer below on load),
        # as the chain's launches do since round 5
        fin = os.environ.get("FIN") == "1" and cin % 16 == 0
        coef = torch.cat([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1])
        lib = Fsp._lib.load()

        def with_in(fn):
            if not fin:
                return fn
            return lambda: (lib.sv_conv_next_input_norm(coef.data_ptr(), 1), fn())
        pf = rb.plan("fwd", cin, cout)
        pb = rb.plan("bwd", cout, cin)
        tplan = float("nan")
        if pf is not None and not name.startswith("_"):
            tplan = timeit(lambda: Fsp.TablePlan(rb.nbr_out, rb.n_out, rb.K).tiles(pf[2]), 5)
        if mode in ("all", "fwd"):
            if pf is None:
                tf = timeit(lambda: Fsp.gather_gemm(x, rb.nbr_out, wt, rb.n_out))
            else:
                ff = Fsp.fragment_cache.get(w)[0]
                tf = timeit(with_in(lambda: Fsp.gather_gemm_planned(x, pf, ff, rb.n_out, K, cin, cout)))
        if mode in ("all", "bwd"):
            if pb is None:
                td = timeit(lambda: Fsp.gather_gemm(dy, rb.table_for_backward_data(), w, rb.n_in))
            else:
                fb = Fsp.fragment_cache.get(w)[1]
                td = timeit(lambda: Fsp.gather_gemm_planned(dy, pb, fb, rb.n_in, K, cout, cin))
        if mode in ("all", "wgrad"):
            wp = rb.wgrad_plan(cin, cout)                   # equal-pieces plan of the table (None with SEEVCN_WGRAD_PLANNED=0: the chunked kernel)
            tw = timeit(with_in(lambda: Fsp.wgrad(x, rb.nbr_out, dy, K, cin, cout, plan=wp)))
        fl = 2.0 * pairs * cin * cout
        print(f"{name:8s} {cin:3d}->{cout:3d} N_in={rb.n_in:7d} N_out={rb.n_out:7d} pairs={pairs:8d} rulebook {tb:7.1f} us plan {tplan:6.1f} us | "
              f"fwd {tf:7.1f} us ({fl / tf / 1e6:6.2f} TF) | bwd-data {td:7.1f} us ({fl / td / 1e6:6.2f} TF) | wgrad {tw:7.1f} us ({fl / tw / 1e6:6.2f} TF)")
        tot += tf + td + tw
        c, shape = rb.out_indices, rb.out_shape
    print(f"sum fwd+bwd+wgrad (one conv per rulebook): {tot:.1f} us")


if __name__ == "__main__":
    main()
