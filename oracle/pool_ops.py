"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement (numpy float32 arithmetic, python loops: small cases only) of CUDA-only reference kernels:
  roiaware_pool3d_forward / backward   <- detector3d/pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-312
  points_in_boxes_cpu                  <- roiaware_pool3d/src/roiaware_pool3d.cpp:121-165
  voxel_query                          <- pointnet2/pointnet2_stack/src/voxel_query_gpu.cu:11-87
  vector_pool / vector_pool_grad       <- pointnet2_stack/src/vector_pool_gpu.cu:217-455
  query_stacked_local_neighbor_idxs    <- vector_pool_gpu.cu:113-190
  query_three_nn_by_stacked_local_idxs <- vector_pool_gpu.cu:18-83
PARITY UNPINNED at op level: the kernels are CUDA-only (cannot run here) and the reference holds no tests for them; restated line
by line.  Rows the reference hands out through a global atomic counter (grouped_idxs, start offsets) come back in a canonical
order here; tests compare them as sets / through the offsets.
"""
import numpy as np

f32 = np.float32


def _local(pt, box, margin):
    x, y, z = f32(pt[0]), f32(pt[1]), f32(pt[2])
    cx, cy, cz, dx, dy, dz, rz = [f32(v) for v in box[:7]]
    if abs(f32(z - cz)) > float(dz) / 2.0:
        return False, f32(0), f32(0)
    cosa, sina = f32(np.cos(-rz)), f32(np.sin(-rz))
    sx, sy = f32(x - cx), f32(y - cy)
    lx = f32(f32(sx * cosa) + f32(sy * f32(-sina)))
    ly = f32(f32(sx * sina) + f32(sy * cosa))
    inside = (abs(float(lx)) < float(dx) / 2.0 + float(f32(margin))) and (abs(float(ly)) < float(dy) / 2.0 + float(f32(margin)))
    return inside, lx, ly


def points_in_boxes_cpu(points, boxes):
    out = np.zeros((len(boxes), len(points)), np.int32)
    for i, b in enumerate(boxes):
        for j, p in enumerate(points):
            out[i, j] = int(_local(p, b, 1e-2)[0])
    return out


def roiaware_pool3d_forward(rois, pts, feat, out_size, max_pts, method):
    ox, oy, oz = out_size
    N, M, C = len(rois), len(pts), feat.shape[1]
    pts_idx = np.zeros((N, ox, oy, oz, max_pts), np.int32)
    argmax = np.zeros((N, ox, oy, oz, C), np.int32)
    pooled = np.zeros((N, ox, oy, oz, C), np.float32)
    for n, b in enumerate(rois):
        dx, dy, dz = f32(b[3]), f32(b[4]), f32(b[5])
        for k, p in enumerate(pts):
            inside, lx, ly = _local(p, b, 1e-5)
            if not inside:
                continue
            lz = f32(f32(p[2]) - f32(b[2]))
            xr, yr, zr = f32(dx / f32(ox)), f32(dy / f32(oy)), f32(dz / f32(oz))
            xi = int(f32(f32(lx + f32(dx / f32(2))) / xr))
            yi = int(f32(f32(ly + f32(dy / f32(2))) / yr))
            zi = int(f32(f32(lz + f32(dz / f32(2))) / zr))
            xi, yi, zi = min(max(xi, 0), ox - 1), min(max(yi, 0), oy - 1), min(max(zi, 0), oz - 1)
            cnt = pts_idx[n, xi, yi, zi, 0]
            if cnt < max_pts - 1:
                pts_idx[n, xi, yi, zi, cnt + 1] = k
                pts_idx[n, xi, yi, zi, 0] += 1
        for v in np.ndindex(ox, oy, oz):
            cnt = pts_idx[n][v][0]
            ids = pts_idx[n][v][1:1 + cnt]
            for c in range(C):
                if method == 0:
                    am, best = -1, -np.inf
                    for i in ids:
                        if feat[i, c] > best:
                            best, am = feat[i, c], i
                    argmax[n][v][c] = am
                    if am != -1:
                        pooled[n][v][c] = best
                else:
                    s = f32(0)
                    for i in ids:
                        s = f32(s + feat[i, c])
                    if cnt > 0:
                        pooled[n][v][c] = f32(s / f32(cnt))
    return pooled, argmax, pts_idx


def roiaware_pool3d_backward(pts_idx, argmax, grad_out, num_pts, method):
    C = grad_out.shape[-1]
    grad_in = np.zeros((num_pts, C), np.float64)
    flat_idx = pts_idx.reshape(-1, pts_idx.shape[-1])
    go = grad_out.reshape(-1, C)
    am = argmax.reshape(-1, C)
    for v in range(len(go)):
        for c in range(C):
            if method == 0:
                if am[v, c] != -1:
                    grad_in[am[v, c], c] += go[v, c]
            else:
                cnt = flat_idx[v, 0]
                g = f32(1) / max(f32(cnt), f32(1))
                for k in range(1, cnt + 1):
                    grad_in[flat_idx[v, k], c] += f32(go[v, c] * g)
    return grad_in.astype(np.float32)


def voxel_query(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    M = len(new_coords)
    B, R1, R2, R3 = point_indices.shape
    idx = np.zeros((M, nsample), np.int32)
    r2 = f32(f32(radius) * f32(radius))
    zr, yr, xr = max_range
    for q in range(M):
        b, cz, cy, cx = [int(v) for v in new_coords[q]]
        cnt = 0
        for dz in range(-zr, zr + 1):
            z = cz + dz
            if z < 0 or z >= R1:
                continue
            for dy in range(-yr, yr + 1):
                y = cy + dy
                if y < 0 or y >= R2:
                    continue
                for dx in range(-xr, xr + 1):
                    x = cx + dx
                    if x < 0 or x >= R3:
                        continue
                    nb = point_indices[b, z, y, x]
                    if nb < 0:
                        continue
                    d = (xyz[nb] - new_xyz[q]).astype(np.float32)
                    d2 = f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))
                    if d2 > r2:
                        continue
                    if cnt < nsample:
                        if cnt == 0:
                            idx[q, :] = nb
                        idx[q, cnt] = nb
                        cnt += 1
        if cnt == 0:
            idx[q, 0] = -1
    return idx


def _in_range(l, dist, neighbor_type):
    if neighbor_type == 1:
        return not (f32(f32(f32(l[0] * l[0]) + f32(l[1] * l[1])) + f32(l[2] * l[2])) > f32(f32(dist) * f32(dist)))
    return not (abs(l[0]) > f32(dist) or abs(l[1]) > f32(dist) or abs(l[2]) > f32(dist))


def _batch_of(q, new_cnt, xyz_cnt):
    bs = int(np.searchsorted(np.cumsum(new_cnt), q, side="right"))
    return bs, int(np.sum(xyz_cnt[:bs])), int(xyz_cnt[bs])


def vector_pool(support_xyz, xyz_cnt, support_features, new_xyz, new_cnt, grid, dist, c_each, use_xyz, nsample, neighbor_type, pooling_type):
    """Returns new_features (M,c_out) RAW sums, new_local_xyz (M,3G) raw sums, point_cnt_of_grid (M,G), grouped rows (R,3)."""
    gx, gy, gz = grid
    G = gx * gy * gz
    M, c_in = len(new_xyz), support_features.shape[1]
    out = np.zeros((M, G * c_each), np.float32)
    oxyz = np.zeros((M, 3 * G), np.float32)
    cnt = np.zeros((M, G), np.int32)
    rows = []
    sx, sy, sz = f32(f32(dist) * 2 / gx), f32(f32(dist) * 2 / gy), f32(f32(dist) * 2 / gz)
    for q in range(M):
        bs, start, n = _batch_of(q, new_cnt, xyz_cnt)
        sample_cnt = 0
        for k in range(n):
            l = (support_xyz[start + k] - new_xyz[q]).astype(np.float32)
            if not _in_range(l, dist, neighbor_type):
                continue
            ix = int(np.floor(f32(f32(l[0] + f32(dist)) / sx)))
            iy = int(np.floor(f32(f32(l[1] + f32(dist)) / sy)))
            iz = int(np.floor(f32(f32(l[2] + f32(dist)) / sz)))
            g = min(max(ix * gy * gz + iy * gz + iz, 0), G - 1)
            if pooling_type == 0:
                cnt[q, g] += 1
                for i in range(c_in):
                    out[q, g * c_each + i % c_each] = f32(out[q, g * c_each + i % c_each] + support_features[start + k, i])
                if use_xyz:
                    oxyz[q, g * 3:g * 3 + 3] = (oxyz[q, g * 3:g * 3 + 3] + l).astype(np.float32)
                rows.append((start + k, q, g))
                sample_cnt += 1
                if nsample > 0 and sample_cnt >= nsample:
                    break
            elif cnt[q, g] == 0:
                cnt[q, g] += 1
                for i in range(c_in):
                    out[q, g * c_each + i % c_each] = support_features[start + k, i]
                if use_xyz:
                    oxyz[q, g * 3:g * 3 + 3] = l
                rows.append((start + k, q, g))
                sample_cnt += 1
                if (nsample > 0 and sample_cnt >= nsample) or sample_cnt >= G:
                    break
    return out, oxyz, cnt, np.array(rows, np.int32).reshape(-1, 3)


def vector_pool_grad(grad_new, cnt, rows, n_support, c_in, c_each):
    g = np.zeros((n_support, c_in), np.float64)
    for sup, q, grid in rows:
        w = f32(1) / max(f32(cnt[q, grid]), f32(1))
        for c in range(c_in):
            g[sup, c] += f32(grad_new[q, grid * c_each + c % c_each] * w)
    return g.astype(np.float32)


def query_stacked_local_neighbor_idxs(support_xyz, xyz_cnt, new_xyz, new_cnt, dist, nsample, neighbor_type):
    """Per query the list of neighbour rows (global), capped at 1000 / nsample."""
    lists = []
    for q in range(len(new_xyz)):
        bs, start, n = _batch_of(q, new_cnt, xyz_cnt)
        cur = []
        for k in range(n):
            l = (support_xyz[start + k] - new_xyz[q]).astype(np.float32)
            if not _in_range(l, dist, neighbor_type):
                continue
            if len(cur) < 1000:
                cur.append(start + k)
            else:
                break
            if nsample > 0 and len(cur) >= nsample:
                break
        lists.append(np.array(cur, np.int32))
    return lists


def query_three_nn_by_stacked_local_idxs(support_xyz, centers, lists):
    M, G = centers.shape[:2]
    idx = np.full((M, G, 3), -1, np.int32)
    d2 = np.zeros((M, G, 3), np.float32)
    for q in range(M):
        for g in range(G):
            b = [1e40, 1e40, 1e40]
            bi = [-1, -1, -1]
            c = centers[q, g]
            for p in lists[q]:
                t = (c - support_xyz[p]).astype(np.float32)
                d = float(f32(f32(f32(t[0] * t[0]) + f32(t[1] * t[1])) + f32(t[2] * t[2])))
                if d < b[0]:
                    b, bi = [d, b[0], b[1]], [p, bi[0], bi[1]]
                elif d < b[1]:
                    b, bi = [b[0], d, b[1]], [bi[0], p, bi[1]]
                elif d < b[2]:
                    b[2], bi[2] = d, p
            if bi[1] == -1:
                bi[1], b[1] = bi[0], b[0]
            if bi[2] == -1:
                bi[2], b[2] = bi[0], b[0]
            idx[q, g] = bi
            d2[q, g] = np.array(b, np.float64).astype(np.float32) if bi[0] != -1 else np.array([np.inf] * 3, np.float32)
    return d2, idx
