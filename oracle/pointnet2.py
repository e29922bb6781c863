"""Oracle: PointNet++ stacked-batch primitives (numpy).  Test infrastructure only.

PARITY UNPINNED by reference tests (there are none) and the CUDA kernels cannot run here; the functions
below restate the kernels line by line, including the thread/tree structure that fixes FPS tie-breaking.
fp32 arithmetic is written without fused multiply-add (the reference's nvcc build may contract a*a+b*b into FMAs;
that only matters for points within one ulp of a radius or of a tie)."""
import numpy as np

F = np.float32


def farthest_point_sampling(xyz, m):
    """farthest_point_sampling_kernel, ops/pointnet2/pointnet2_stack/src/sampling_gpu.cu:24-140 (one scene: xyz (n,3))."""
    xyz = np.asarray(xyz, F)
    n = len(xyz)
    t = max(min(1 << int(np.log(n) / np.log(2.0)), 1024), 1)       # opt_n_threads, :9-13
    temp = np.full(n, 1e10, F)                                     # pointnet2_utils.py:178
    idxs = np.zeros(m, np.int32)
    old = 0                                                         # :44-46
    rows = -(-n // t)
    pad = rows * t - n
    for j in range(1, m):
        d = xyz - xyz[old]
        dist = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(F) + (d[:, 2] * d[:, 2]).astype(F)   # :62
        d2 = np.minimum(dist, temp)                                 # :63 (min returns the non-NaN operand)
        temp = d2
        # per-thread strided scan with strict '>' (:55-67): first maximum of each residue class k mod t
        grid = np.concatenate([d2, np.full(pad, -np.inf, F)]).reshape(rows, t)
        arg = np.argmax(grid, axis=0)
        vals = grid[arg, np.arange(t)].copy()
        inds = (arg * t + np.arange(t)).astype(np.int64)
        best_init = vals > -1                                        # best = -1, besti = 0 (:51-52)
        vals = np.where(best_init, vals, F(-1))
        inds = np.where(best_init, inds, 0)
        half = t // 2
        while half >= 1:                                             # tree with "keep idx1 unless strictly greater" (:16-21)
            v1, v2 = vals[:half], vals[half:2 * half]
            take = v2 > v1
            vals[:half] = np.maximum(v1, v2)
            inds[:half] = np.where(take, inds[half:2 * half], inds[:half])
            half //= 2
        old = int(inds[0])
        idxs[j] = old
    return idxs


def ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
    """ball_query_kernel_stack, src/ball_query_gpu.cu:16-66 -> idx (M, nsample) int32 (raw kernel output: -1 marks an empty ball)."""
    xyz, new_xyz = np.asarray(xyz, F), np.asarray(new_xyz, F)
    r2 = F(radius) * F(radius)
    idx = np.zeros((len(new_xyz), nsample), np.int32)
    ps = np.concatenate([[0], np.cumsum(xyz_batch_cnt)])
    qs = np.concatenate([[0], np.cumsum(new_xyz_batch_cnt)])
    for b in range(len(xyz_batch_cnt)):
        pts = xyz[ps[b]:ps[b + 1]]
        for q in range(qs[b], qs[b + 1]):
            d = new_xyz[q] - pts
            d2 = ((d[:, 0] * d[:, 0]).astype(F) + (d[:, 1] * d[:, 1]).astype(F)).astype(F) + (d[:, 2] * d[:, 2]).astype(F)
            hits = np.nonzero(d2 < r2)[0][:nsample]
            if len(hits) == 0:
                idx[q, 0] = -1                                       # :65
            else:
                idx[q, :] = hits[0]                                  # :55-59
                idx[q, :len(hits)] = hits
    return idx


def group_points(features, features_batch_cnt, idx, idx_batch_cnt):
    """group_points_kernel_stack, src/group_points_gpu.cu:71-102 -> (M, C, nsample)."""
    fs = np.concatenate([[0], np.cumsum(features_batch_cnt)])[:-1]
    row_start = np.repeat(fs, idx_batch_cnt)
    return np.transpose(np.asarray(features)[row_start[:, None] + idx], (0, 2, 1)).copy()


def group_points_grad(grad_out, idx, idx_batch_cnt, features_batch_cnt, n):
    """group_points_grad_kernel_stack, :15-45 (scatter-add)."""
    fs = np.concatenate([[0], np.cumsum(features_batch_cnt)])[:-1]
    row_start = np.repeat(fs, idx_batch_cnt)
    g = np.zeros((n, grad_out.shape[1]), np.float64)
    rows = (row_start[:, None] + idx)                                # (M, ns)
    np.add.at(g, rows.reshape(-1), np.transpose(grad_out, (0, 2, 1)).reshape(-1, grad_out.shape[1]))
    return g


# ------------------------------------------------------------------------------------------ batch layout + 3-NN interpolation
def ball_query_batch(radius, nsample, xyz, new_xyz):
    """pointnet2_batch/src/ball_query_gpu.cu:13-48.  xyz (B,N,3), new_xyz (B,M,3) -> idx (B,M,nsample) int32 (zeros when no hit)."""
    B, M = new_xyz.shape[:2]
    idx = np.zeros((B, M, nsample), np.int32)
    r2 = F(radius) * F(radius)
    for b in range(B):
        for q in range(M):
            d = new_xyz[b, q].astype(F) - xyz[b].astype(F)
            d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
            hit = np.nonzero(d2 < r2)[0][:nsample]
            if len(hit):
                idx[b, q, :] = hit[0]
                idx[b, q, :len(hit)] = hit
    return idx


def three_nn(unknown, known):
    """interpolate_gpu.cu three_nn_kernel: float squared distances, strict '<' cascade -> (dist2 (N,3) float32, idx (N,3))."""
    u, k = np.asarray(unknown, F), np.asarray(known, F)
    dist2, idx = np.empty((len(u), 3), F), np.zeros((len(u), 3), np.int32)
    for i in range(len(u)):
        d = u[i] - k
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(F)
        order = np.argsort(d2, kind='stable')[:3]                  # first-come wins among equal distances, like the cascade
        best = np.full(3, 1e40)
        bi = np.zeros(3, np.int32)
        best[:len(order)] = d2[order]
        bi[:len(order)] = order
        dist2[i], idx[i] = best.astype(F), bi
    return dist2, idx


def three_interpolate(features_mc, idx, weight):
    """stack layout: features (M,C) -> (N,C) = w0 f[i0] + w1 f[i1] + w2 f[i2] in fp32, left to right."""
    f, w = np.asarray(features_mc, F), np.asarray(weight, F)
    return (w[:, 0:1] * f[idx[:, 0]] + w[:, 1:2] * f[idx[:, 1]]) + w[:, 2:3] * f[idx[:, 2]]


def sa_scale_train(xyz, features, new_xyz, idx, row_start, w1, g1, b1, w2, g2, b2, eps=1e-5):
    """One radius scale of StackSAModuleMSG.forward in TRAINING mode (pointnet2_modules.py:96-110 on top of QueryAndGroup,
    pointnet2_utils.py:112-159), in float64: grouped [xyz - new_xyz | features] with all-zero groups for empty balls (idx[m][0] < 0, as
    ball_query leaves them), Conv2d 1x1 (no bias) -> BatchNorm2d with BATCH statistics over all M * nsample positions (biased variance) ->
    ReLU, twice, then the maximum over nsample.  idx (M, ns) scene-local, row_start (M,) first support row of the query's scene;
    w1 (C1, 3 + C), w2 (C2, C1).  -> (M, C2) float64."""
    M, ns = idx.shape
    empty = idx[:, 0] < 0
    rows = row_start[:, None].astype(np.int64) + np.where(empty[:, None], 0, idx)
    x = xyz[rows].astype(np.float64) - new_xyz[:, None, :].astype(np.float64)
    if features is not None:
        x = np.concatenate([x, features[rows].astype(np.float64)], axis=2)
    x[empty] = 0.0
    x = x.reshape(M * ns, -1)

    def bn_relu(z, g, b):
        return np.maximum((z - z.mean(0)) / np.sqrt(z.var(0) + eps) * g.astype(np.float64) + b.astype(np.float64), 0.0)

    a1 = bn_relu(x @ w1.astype(np.float64).T, g1, b1)
    a2 = bn_relu(a1 @ w2.astype(np.float64).T, g2, b2)
    return a2.reshape(M, ns, -1).max(axis=1)
