#!/usr/bin/env python3
"""Per-kernel roofline table at bench-workload sizes: time (HIP events, warm), algorithmic bytes per SURVEY.md 8(d) / DESIGN.md 3,
achieved GB/s against the 8 TB/s HBM peak (or TFLOP/s against the 157.3 TFLOP/s fp32-MFMA peak).  Prints markdown."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import seevcn_amd.synth as synth
from seevcn_amd.pcdet.ops import voxel_ops
from seevcn_amd.pcdet.ops.iou3d_nms import iou3d_nms_utils
from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as PS
from seevcn_amd.spconv import functional as Fsp, norm
from seevcn_amd.vcn.utils import sampling
from seevcn_amd.vcn.scene_merge import complete_scene_batch_device

HBM, MFMA = 8000.0, 157.3
rows = []


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3  # us


def row(name, us, nbytes=None, flops=None, note=""):
    gbs = nbytes / us / 1e3 if nbytes else None
    tf = flops / us / 1e6 if flops else None
    frac = (gbs / HBM) if gbs is not None and flops is None else (tf / MFMA if tf is not None else None)
    rows.append((name, us, nbytes, gbs, tf, frac, note))


def main():
    dev = torch.device("cuda:0")
    bs = 16
    pts, gt = synth.make_scene_batch(bs, seed=2000)
    g = dict(r=[0, -40, -3, 70.4, 40, 1], v=[0.05, 0.05, 0.1], grid=[1408, 1600, 40])
    p = torch.from_numpy(pts).to(dev)
    P = p.shape[0]
    feats, coords, _ = voxel_ops.voxelize_dynamic(p, g["r"], g["v"], g["grid"], bs)
    V = coords.shape[0]
    row("dynamic voxelise (3 kernels + scan)", timeit(lambda: voxel_ops.voxelize_dynamic(p, g["r"], g["v"], g["grid"], bs)), 16 * P + 28 * V, note=f"P={P}, V={V}; includes one host sync (.item)")
    # hard voxeliser (PointPillars geometry)
    cnt = torch.bincount(p[:, 0].long(), minlength=bs).int()
    pr, pv, pg = [0, -39.68, -3, 69.12, 39.68, 1], [0.16, 0.16, 4], [432, 496, 1]
    order = torch.argsort(p[:, 0], stable=True)
    ps = p[order].contiguous()
    hv = voxel_ops.voxelize_hard(ps, 1, 3, cnt, pr, pv, pg, 32, 16000)
    nv = int(hv[3].sum().item()) if isinstance(hv, tuple) and len(hv) > 3 else 0
    row("hard voxelise (1 WG / scene)", timeit(lambda: voxel_ops.voxelize_hard(ps, 1, 3, cnt, pr, pv, pg, 32, 16000), 5), 12 * P + (4 * 32 * 3 + 20) * max(nv, 1), note=f"{bs} scenes, V={nv}, 32 pts/voxel")
    # BatchNorm + ReLU on the conv3-level feature matrix
    import torch.nn as nn
    N, C = 134580, 64
    x = torch.randn(N, C, device=dev, requires_grad=True)
    bn = nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev).train()
    y = norm.batch_norm_relu(bn, x, True)
    dy = torch.randn_like(y)
    row("BatchNorm+ReLU forward (3 kernels)", timeit(lambda: norm.batch_norm_relu(bn, x, True)), 4 * N * C * 3, note=f"N={N}, C={C}")
    from seevcn_amd import _lib
    lib = _lib.load()
    xd = x.detach()
    mean, invstd = xd.mean(0).contiguous(), (1.0 / torch.sqrt(xd.var(0, unbiased=False) + 1e-3)).contiguous()
    dx, dg, db = torch.empty_like(xd), torch.empty(C, device=dev), torch.empty(C, device=dev)
    scr = _lib.workspace.scratch(f"bn{C}", lib.sv_batchnorm_scratch_bytes(C), dev)

    def bn_bwd():
        _lib.check(lib.sv_batchnorm_relu_backward(_lib.ptr(xd), _lib.ptr(dy), N, C, _lib.ptr(bn.weight), _lib.ptr(bn.bias), _lib.ptr(mean), _lib.ptr(invstd), 1,
                                                  _lib.ptr(scr), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), _lib.stream()), "bn bwd")
    row("BatchNorm+ReLU backward (3 kernels)", timeit(bn_bwd), 4 * N * C * 5, note="x, dy read twice, dx written")
    # rulebooks at the conv1 level (213k voxels)
    shape = [41, 1600, 1408]
    rb = Fsp.build_subm_rulebook(coords, bs, shape, [3, 3, 3])
    pairs = int(rb.pair_counts().sum().item())
    row("submanifold rulebook 3x3x3", timeit(lambda: Fsp.build_subm_rulebook(coords, bs, shape, [3, 3, 3]), 5), 16 * V + 4 * 27 * V + 16 * V, note=f"N={V}, pairs={pairs}")
    rs = Fsp.build_sparse_rulebook(coords, bs, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    row("strided rulebook 3x3x3 s2", timeit(lambda: Fsp.build_sparse_rulebook(coords, bs, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1]), 5), 16 * V + 2 * 4 * 27 * V + 16 * rs.n_out, note=f"N_in={V}, N_out={rs.n_out}; includes one host sync")
    # dense (HeightCompression) at the backbone output
    No, Co = 57955, 128
    oc = torch.unique(torch.stack([torch.randint(0, bs, (No * 2,), device=dev), torch.randint(0, 2, (No * 2,), device=dev), torch.randint(0, 200, (No * 2,), device=dev),
                                   torch.randint(0, 176, (No * 2,), device=dev)], 1).int(), dim=0)[:No].contiguous()
    of = torch.randn(oc.shape[0], Co, device=dev)
    row(".dense() (HeightCompression)", timeit(lambda: Fsp.sparse_to_dense(of, oc, bs, [2, 200, 176])), 4 * oc.shape[0] * Co + 4 * bs * Co * 2 * 200 * 176, note=f"{bs}x{Co}x2x200x176 = {4 * bs * Co * 2 * 200 * 176 / 1e6:.0f} MB written once")
    # FPS + ball query + NMS (PV-RCNN shapes)
    one = p[p[:, 0] == 0][:, 1:4].contiguous()
    n1 = one.shape[0]
    row("FPS 2048 of one scene", timeit(lambda: PS.farthest_point_sample(one.unsqueeze(0), 2048), 5), 16 * n1 + 4 * 2048, note=f"N={n1}; latency-bound: 2047 dependent rounds ({timeit(lambda: PS.farthest_point_sample(one.unsqueeze(0), 2048), 3) / 2047:.2f} us / round)")
    boxes = torch.from_numpy(np.concatenate([gt[0][gt[0, :, 3] > 0][:, :7].repeat(150, 0) + np.random.default_rng(0).normal(0, 0.3, (gt[0][gt[0, :, 3] > 0].shape[0] * 150, 7))], 0).astype(np.float32)).to(dev)[:9000].contiguous()
    scores = torch.rand(boxes.shape[0], device=dev)
    row("rotated NMS 9000 boxes (mask + sweep)", timeit(lambda: iou3d_nms_utils.nms_gpu(boxes, scores, 0.8), 5), None, note=f"{boxes.shape[0] ** 2 / 2 / 1e6:.1f} M IoU pairs; greedy sweep on the GPU, no D2H of the mask")
    # VCN post-processing (64 objects)
    from post_inputs import make_pairs
    partial, coarse = make_pairs(64, seed=5)
    pp, cc = torch.from_numpy(partial).to(dev), torch.from_numpy(coarse).to(dev)
    row("surface select k=30, 64 objects", timeit(lambda: sampling.get_partial_mesh_batch_device(pp, cc, k=30)), 64 * 36864, note="latency-bound (sort, k wave-min rounds, set replay)")
    surf, _ = sampling.get_partial_mesh_batch_device(pp, cc, k=30)
    row("largest DBSCAN cluster, 64 objects", timeit(lambda: sampling.get_largest_cluster_batch_device(surf, eps=0.4, min_points=2)), 64 * 24576, note="VALU (fp64 distance tests), includes one host sync")
    print("| kernel group | time (us) | algorithmic bytes | achieved | fraction of peak | note |")
    print("|---|---|---|---|---|---|")
    for name, us, nb, gbs, tf, frac, note in rows:
        ach = f"{gbs:.0f} GB/s" if gbs is not None and tf is None else (f"{tf:.1f} TFLOP/s" if tf is not None else "—")
        print(f"| {name} | {us:.1f} | {nb / 1e6:.1f} MB |" if nb else f"| {name} | {us:.1f} | — |", ach, "|", f"{frac * 100:.1f} %" if frac is not None else "—", "|", note, "|")


if __name__ == "__main__":
    main()
