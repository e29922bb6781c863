"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's VCN post-processing, which the reference itself runs on the CPU after the network
(see/surface_completion/models/VCN.py:89-93):

  partial_with_kdtree   <- models/vcn/utils/sampling.py:8-41   (np.unique -> cKDTree k-NN -> list(set()) -> np.tile)
  largest_cluster       <- models/vcn/utils/sampling.py:83-100 (open3d cluster_dbscan -> bincount argmax -> np.tile)
  merge_instances       <- SEE_VCN.py:115,244                  (np.unique(np.vstack(clustered), axis=0))
  replace_with_completed_pts <- SEE_VCN.py:247-265             (open3d compute_point_cloud_distance < 0.1 -> vstack)

Pinning: partial_with_kdtree is pinned bit-exactly against tests/golden/vcn_post.npz, produced by the reference's own
get_partial_mesh_batch (scipy cKDTree + CPython set) in the build container (tests/golden/make_post_golden.py).
largest_cluster / replace_with_completed_pts call open3d, which is NOT installed here and not vendored under /root/reference
(third-party, `open3d` pip package, no version pin in docker/Dockerfile): PARITY UNPINNED w.r.t. open3d.  The restatement follows
open3d's published ClusterDBSCAN (cpp/open3d/geometry/PointCloudCluster.cpp: radius search with strict d^2 < eps^2 in
float64 via nanoflann, sequential label assignment in point order, BFS expansion through core points) and is cross-checked in
tests against sklearn.cluster.DBSCAN (largest-cluster membership is label-invariant).
"""
import numpy as np

LINEAR_PROBES, PERTURB_SHIFT = 9, 5


def cpython_set_order(values):
    """Iteration order of `set(values)` for non-negative ints under CPython 3.10 (Objects/setobject.c: set_add_entry,
    set_insert_clean, set_table_resize).  Kept as the readable statement of what the HIP kernel emulates; the restatement
    below uses the interpreter's own set, and tests check the two agree."""
    def insert_clean(table, mask, h):
        perturb, i = h, h & mask
        while True:
            if table[i] < 0:
                table[i] = h
                return
            if i + LINEAR_PROBES <= mask:
                for j in range(1, LINEAR_PROBES + 1):
                    if table[i + j] < 0:
                        table[i + j] = h
                        return
            perturb >>= PERTURB_SHIFT
            i = (i * 5 + 1 + perturb) & mask

    mask, table, fill = 7, [-1] * 8, 0
    for h in values:
        h = int(h)
        perturb, i = h, h & mask
        done = False
        while not done:
            probes = LINEAR_PROBES if i + LINEAR_PROBES <= mask else 0
            for j in range(probes + 1):
                e = table[i + j]
                if e == h:
                    done = True
                    break
                if e < 0:
                    table[i + j] = h
                    fill += 1
                    if fill * 5 >= mask * 3:
                        newsize = 8
                        while newsize <= fill * 4:
                            newsize <<= 1
                        old, table, mask = table, [-1] * newsize, newsize - 1
                        for o in old:
                            if o >= 0:
                                insert_clean(table, mask, o)
                    done = True
                    break
            if not done:
                perturb >>= PERTURB_SHIFT
                i = (i * 5 + 1 + perturb) & mask
    return [e for e in table if e >= 0]


def knn_indices(queries, complete, k):
    """k nearest rows of `complete` per query, ascending float64 squared distance (ties: lower index), like cKDTree.query."""
    q = queries.astype(np.float64)
    c = complete.astype(np.float64)
    out = np.empty((len(q), k), np.int64)
    for i in range(len(q)):
        d = c - q[i]
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        out[i] = np.argsort(d2, kind='stable')[:k]
    return out


def partial_with_kdtree(partial_pc, complete_pc, k, surface_pts=1024):
    """sampling.py:8-41.  Returns (surface (surface_pts,3), n_selected)."""
    partial = np.unique(np.asarray(partial_pc), axis=0)
    complete = np.asarray(complete_pc)
    surface_idx = []
    for row in knn_indices(partial, complete, k):
        surface_idx.extend(row)                      # np.int64 items, as in the reference
    surface_idx = list(set(surface_idx))             # CPython set order
    sel = complete[surface_idx]
    return np.tile(sel, [surface_pts, 1])[:surface_pts, :], len(surface_idx)


def get_partial_mesh_batch(batch_partial, batch_complete, k=20, surface_pts=1024):
    return np.stack([partial_with_kdtree(p, c, k, surface_pts)[0] for p, c in zip(batch_partial, batch_complete)])


def dbscan_labels(pc, eps, min_points):
    """open3d PointCloud.cluster_dbscan semantics (see module docstring)."""
    x = np.asarray(pc, np.float64)
    n = len(x)
    eps2 = float(eps) * float(eps)
    nbs = []
    for i in range(n):
        d = x - x[i]
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        nbs.append(np.nonzero(d2 < eps2)[0])
    labels = np.full(n, -2, np.int64)
    cluster = 0
    for idx in range(n):
        if labels[idx] != -2:
            continue
        if len(nbs[idx]) < min_points:
            labels[idx] = -1
            continue
        nxt, visited = set(nbs[idx].tolist()), {idx}
        labels[idx] = cluster
        while nxt:
            nb = nxt.pop()
            visited.add(nb)
            if labels[nb] == -1:
                labels[nb] = cluster
            if labels[nb] != -2:
                continue
            labels[nb] = cluster
            if len(nbs[nb]) >= min_points:
                nxt.update(q for q in nbs[nb].tolist() if q not in visited)
        cluster += 1
    return labels


def largest_cluster(pc, eps=0.4, min_points=1, total_pts=1024):
    """sampling.py:83-100.  Returns (points (total_pts,3) float64, cluster size); raises ValueError when all points are noise
    (np.argmax of an empty bincount, as the reference would)."""
    pc = np.asarray(pc)
    labels = dbscan_labels(pc, eps, min_points)
    y = np.bincount(labels[labels >= 0])
    value = np.argmax(y)
    members = np.argwhere(labels == value)[:, 0]
    ret = pc[members].astype(np.float64)
    return np.tile(ret, (int(np.ceil(total_pts / ret.shape[0])), 1))[:total_pts, :], len(members)


def get_largest_cluster_batch(pc, eps=0.4, min_points=1, total_pts=1024):
    return np.stack([largest_cluster(p, eps, min_points, total_pts)[0] for p in pc])


def merge_instances(clustered):
    return np.unique(np.vstack(clustered), axis=0)


def replace_with_completed_pts(points, sc_instances, point_dist_thresh=0.1, chunk=2048):
    """SEE_VCN.py:247-265 on plain arrays: drop scene points closer than the threshold to any completed point, prepend the
    completed points."""
    if sc_instances is None:
        return np.asarray(points)
    p = np.asarray(points, np.float64)
    r = np.asarray(sc_instances, np.float64)
    near = np.zeros(len(p), bool)
    for s in range(0, len(p), chunk):
        d = p[s:s + chunk, None, :] - r[None, :, :]
        d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
        near[s:s + chunk] = np.sqrt(d2.min(axis=1)) < point_dist_thresh
    return np.vstack((r, p[~near])), near
