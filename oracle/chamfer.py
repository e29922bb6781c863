"""ORACLE (test infrastructure only): numpy restatement of the reference's Chamfer CUDA extension
(see/surface_completion/models/vcn/extensions/chamfer_dist/chamfer.cu:15-201).  The kernels are CUDA-only and the reference
has no test for them: PARITY UNPINNED at op level; the restatement follows the kernel line by line (fp32, no FMA contraction,
first minimum wins) and is cross-checked against scipy.spatial.cKDTree in tests/test_chamfer.py."""
import numpy as np

F = np.float32


def nn_dist(xyz1, xyz2):
    """chamfer_dist_kernel: per point of xyz1 (B,n,3) the squared distance to and index of its nearest point of xyz2 (B,m,3)."""
    a, b = np.asarray(xyz1, F), np.asarray(xyz2, F)
    B, n, _ = a.shape
    dist, idx = np.empty((B, n), F), np.empty((B, n), np.int32)
    for i in range(B):
        dx = b[i, None, :, 0] - a[i, :, None, 0]
        dy = b[i, None, :, 1] - a[i, :, None, 1]
        dz = b[i, None, :, 2] - a[i, :, None, 2]
        d = (dx * dx + dy * dy) + dz * dz                    # fp32, left to right
        idx[i] = np.argmin(d, axis=1)                         # first minimum
        dist[i] = d[np.arange(n), idx[i]]
    return dist, idx


def forward(xyz1, xyz2):
    d1, i1 = nn_dist(xyz1, xyz2)
    d2, i2 = nn_dist(xyz2, xyz1)
    return d1, d2, i1, i2


def backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2):
    """chamfer_dist_grad_kernel (float64 accumulation here; the kernel's float atomics have no defined order)."""
    a, b = np.asarray(xyz1, np.float64), np.asarray(xyz2, np.float64)
    g1, g2 = np.zeros_like(a), np.zeros_like(b)
    for i in range(a.shape[0]):
        for (p, q, idx, gd, gp, gq) in ((a[i], b[i], idx1[i], grad_dist1[i], g1[i], g2[i]), (b[i], a[i], idx2[i], grad_dist2[i], g2[i], g1[i])):
            g = (2.0 * np.asarray(gd, np.float64))[:, None] * (p - q[idx])
            gp += g
            np.subtract.at(gq, idx, g)
    return g1.astype(F), g2.astype(F)
