"""Oracle: anchor-head arithmetic in numpy (anchors, nearest-BEV IoU target assignment, residual box coding, decode).
Test infrastructure only.  Pinned by tests/golden/second_head.npz (outputs of the reference's AnchorHeadSingle)."""
import numpy as np

F = np.float32
PI = F(np.pi)


def limit_period(val, offset=0.5, period=np.pi):
    """detector3d/pcdet/utils/common_utils.py:22-25 (fp32)."""
    val = np.asarray(val, F)
    return val - np.floor(val / F(period) + F(offset)) * F(period)


def generate_anchors(anchor_cfgs, grid_size, pc_range):
    """AnchorGenerator.generate_anchors, dense_heads/target_assigner/anchor_generator.py:17-60 (align_center False).
    Returns the head-order flat anchors (A,7): [(z,y,x), class set, size, rot] and anchors per location per set.
    np.arange in float32 accumulates like torch.arange (start + i*step in higher precision, rounded)."""
    sets = []
    for c in anchor_cfgs:
        fm = np.asarray(grid_size[:2]) // c['feature_map_stride']
        xs_n, ys_n = int(fm[0]), int(fm[1])
        x_stride = (pc_range[3] - pc_range[0]) / (xs_n - 1)
        y_stride = (pc_range[4] - pc_range[1]) / (ys_n - 1)
        xs = (pc_range[0] + np.arange(xs_n, dtype=np.float64) * x_stride).astype(F)
        ys = (pc_range[1] + np.arange(ys_n, dtype=np.float64) * y_stride).astype(F)
        sizes = np.asarray(c['anchor_sizes'], F).reshape(-1, 3)
        rots = np.asarray(c['anchor_rotations'], F)
        zs = np.asarray(c['anchor_bottom_heights'], F)
        a = np.zeros((len(zs), ys_n, xs_n, len(sizes), len(rots), 7), F)
        a[..., 0] = xs[None, None, :, None, None]
        a[..., 1] = ys[None, :, None, None, None]
        a[..., 2] = zs[:, None, None, None, None]
        a[..., 3:6] = sizes[None, None, None, :, None, :]
        a[..., 6] = rots[None, None, None, None, :]
        a[..., 2] += a[..., 5] / 2
        sets.append(a.reshape(len(zs), ys_n, xs_n, -1, 7))
    per_set = [s.shape[3] for s in sets]
    return np.concatenate(sets, axis=3).reshape(-1, 7), per_set


def nearest_bev(b):
    """boxes3d_lidar_to_aligned_bev_boxes, detector3d/pcdet/utils/box_utils.py:312-323."""
    rot = np.abs(limit_period(b[:, 6], 0.5, np.pi))
    keep = (rot < PI / 4)[:, None]
    dims = np.where(keep, b[:, [3, 4]], b[:, [4, 3]])
    return np.concatenate([b[:, 0:2] - dims / 2, b[:, 0:2] + dims / 2], axis=1)


def iou_normal(a, b):
    """boxes_iou_normal, box_utils.py:286-309."""
    xl = np.clip(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    yl = np.clip(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = xl * yl
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.clip(aa[:, None] + ab[None, :] - inter, F(1e-6), None)


def encode(boxes, anchors):
    """ResidualCoder.encode_torch, detector3d/pcdet/utils/box_coder_utils.py:13-46."""
    ad, gd = np.clip(anchors[:, 3:6], F(1e-5), None), np.clip(boxes[:, 3:6], F(1e-5), None)
    diag = np.sqrt(ad[:, 0] ** 2 + ad[:, 1] ** 2)
    return np.stack([(boxes[:, 0] - anchors[:, 0]) / diag, (boxes[:, 1] - anchors[:, 1]) / diag, (boxes[:, 2] - anchors[:, 2]) / ad[:, 2],
                     np.log(gd[:, 0] / ad[:, 0]), np.log(gd[:, 1] / ad[:, 1]), np.log(gd[:, 2] / ad[:, 2]), boxes[:, 6] - anchors[:, 6]], 1).astype(F)


def decode(enc, anchors):
    """ResidualCoder.decode_torch, box_coder_utils.py:48-77."""
    diag = np.sqrt(anchors[..., 3] ** 2 + anchors[..., 4] ** 2)
    return np.stack([enc[..., 0] * diag + anchors[..., 0], enc[..., 1] * diag + anchors[..., 1], enc[..., 2] * anchors[..., 5] + anchors[..., 2],
                     np.exp(enc[..., 3]) * anchors[..., 3], np.exp(enc[..., 4]) * anchors[..., 4], np.exp(enc[..., 5]) * anchors[..., 5],
                     enc[..., 6] + anchors[..., 6]], -1).astype(F)


def generate_predicted_boxes(enc, anchors, dir_logits, dir_offset, dir_limit_offset, num_bins):
    """anchor_head_template.py:225-272."""
    boxes = decode(enc, anchors[None])
    if dir_logits is not None:
        lab = np.argmax(dir_logits, axis=-1)
        period = 2 * np.pi / num_bins
        rot = limit_period(boxes[..., 6] - F(dir_offset), dir_limit_offset, period)
        boxes[..., 6] = rot + F(dir_offset) + F(period) * lab.astype(F)
    return boxes


def assign_targets(anchors, per_set, set_classes, matched, unmatched, gt_boxes):
    """AxisAlignedTargetAssigner.assign_targets(+_single), axis_aligned_target_assigner.py:36-210, deterministic branch.
    anchors (A,7) head order; gt_boxes (B,G,8). Returns labels (B,A) int32, targets (B,A,7), reg_weights (B,A)."""
    A = len(anchors)
    per_loc = sum(per_set)
    offs = np.concatenate([[0], np.cumsum(per_set)])
    slot = np.arange(A) % per_loc
    B = gt_boxes.shape[0]
    labels = np.zeros((B, A), np.int32)
    targets = np.zeros((B, A, 7), F)
    weights = np.zeros((B, A), F)
    ab = nearest_bev(anchors)
    for b in range(B):
        gt = gt_boxes[b]
        for s in range(len(per_set)):
            idx = np.nonzero((slot >= offs[s]) & (slot < offs[s + 1]))[0]
            g = gt[gt[:, 7].astype(int) == set_classes[s]]
            lab = np.full(len(idx), -1, np.int32)
            if len(g) == 0:
                labels[b, idx] = 0
                continue
            ov = iou_normal(ab[idx], nearest_bev(g[:, :7]))                      # :140-141
            arg = ov.argmax(1)
            mx = ov[np.arange(len(idx)), arg]
            gmax = ov.max(0)
            gmax[gmax == 0] = -1                                                 # :152-153
            force = (ov == gmax[None]).any(1)                                    # :155-158
            lab[force] = set_classes[s]
            pos = mx >= F(matched[s])                                            # :160-163
            lab[pos] = set_classes[s]
            fg = lab > 0
            lab[mx < F(unmatched[s])] = 0                                        # :164, :186
            lab[force] = set_classes[s]                                          # :187
            labels[b, idx] = lab
            t = np.zeros((len(idx), 7), F)
            t[fg] = encode(g[arg[fg], :7], anchors[idx][fg])                     # :189-193
            targets[b, idx] = t
            weights[b, idx] = (lab > 0).astype(F)                                # :195-201
    return labels, targets, weights


# ------------------------------------------------------------------------------------------ CenterHead targets
def gaussian_radius(h, w, min_overlap=0.5):
    """CenterNet radius from box size on the feature map (reference model_utils/centernet_utils.py:10-33)."""
    a1, b1, c1 = 1.0, h + w, w * h * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * a1 * c1)) / 2
    a2, b2, c2 = 4.0, 2 * (h + w), (1 - min_overlap) * w * h
    r2 = (b2 + np.sqrt(b2 ** 2 - 4 * a2 * c2)) / 2
    a3, b3, c3 = 4 * min_overlap, -2 * min_overlap * (h + w), (min_overlap - 1) * w * h
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def center_assign_single_head(num_classes, gt_boxes, fm_size, stride, pc_range, voxel_size, num_max_objs=500, gaussian_overlap=0.1, min_radius=2):
    """One head, one scene (reference dense_heads/center_head.py:112-173 + centernet_utils.py:36-78).
    gt_boxes (N, 8+) with the head-local 1-based class in the last column; fm_size = (W, H)."""
    W, H = fm_size
    D = gt_boxes.shape[1]
    heat = np.zeros((num_classes, H, W), np.float32)
    ret = np.zeros((num_max_objs, D - 1 + 1), np.float32)
    inds = np.zeros(num_max_objs, np.int64)
    mask = np.zeros(num_max_objs, np.int64)
    for k in range(min(num_max_objs, len(gt_boxes))):
        x, y, z, dx, dy = [np.float32(v) for v in gt_boxes[k, :5]]
        cx = np.float32(np.float32(np.float32(x - np.float32(pc_range[0])) / np.float32(voxel_size[0])) / np.float32(stride))
        cy = np.float32(np.float32(np.float32(y - np.float32(pc_range[1])) / np.float32(voxel_size[1])) / np.float32(stride))
        cx = min(max(cx, np.float32(0)), np.float32(W - 0.5))
        cy = min(max(cy, np.float32(0)), np.float32(H - 0.5))
        ix, iy = int(cx), int(cy)
        fdx = np.float32(np.float32(dx / np.float32(voxel_size[0])) / np.float32(stride))
        fdy = np.float32(np.float32(dy / np.float32(voxel_size[1])) / np.float32(stride))
        if fdx <= 0 or fdy <= 0:
            continue
        if not (0 <= ix <= W and 0 <= iy <= H):
            continue
        radius = max(int(np.float32(gaussian_radius(np.float32(fdx), np.float32(fdy), np.float32(gaussian_overlap)))), min_radius)
        cls = int(gt_boxes[k, -1]) - 1
        sigma = (2 * radius + 1) / 6
        left, right, top, bottom = min(ix, radius), min(W - ix, radius + 1), min(iy, radius), min(H - iy, radius + 1)
        yy, xx = np.ogrid[-radius:radius + 1, -radius:radius + 1]
        g = np.exp(-(xx * xx + yy * yy) / (2 * sigma * sigma))
        g[g < np.finfo(g.dtype).eps * g.max()] = 0
        sub = heat[cls, iy - top:iy + bottom, ix - left:ix + right]
        gs = g[radius - top:radius + bottom, radius - left:radius + right].astype(np.float32)
        if min(gs.shape) > 0 and min(sub.shape) > 0:
            np.maximum(sub, gs, out=sub)
        inds[k] = iy * W + ix
        mask[k] = 1
        ret[k, 0:2] = [cx - np.float32(ix), cy - np.float32(iy)]
        ret[k, 2] = z
        ret[k, 3:6] = np.log(gt_boxes[k, 3:6].astype(np.float32))
        ret[k, 6], ret[k, 7] = np.cos(np.float32(gt_boxes[k, 6])), np.sin(np.float32(gt_boxes[k, 6]))
        if D > 8:
            ret[k, 8:] = gt_boxes[k, 7:-1]
    return heat, ret, inds, mask


def center_assign_targets(gt_boxes, class_names, class_names_each_head, fm_size, stride, pc_range, voxel_size, **kw):
    """All heads / scenes (reference center_head.py:175-231): boxes routed to the head holding their class, original order kept."""
    names = ['bg'] + list(class_names)
    out = {'heatmaps': [], 'target_boxes': [], 'inds': [], 'masks': []}
    for head_names in class_names_each_head:
        per = [[], [], [], []]
        for b in range(gt_boxes.shape[0]):
            rows = []
            for box in gt_boxes[b]:
                n = names[int(box[-1])]
                if n in head_names:
                    r = box.copy()
                    r[-1] = head_names.index(n) + 1
                    rows.append(r)
            cur = np.stack(rows) if rows else np.zeros((0, gt_boxes.shape[-1]), np.float32)
            res = center_assign_single_head(len(head_names), cur, fm_size, stride, pc_range, voxel_size, **kw)
            for lst, r in zip(per, res):
                lst.append(r)
        for key, lst in zip(out, per):
            out[key].append(np.stack(lst))
    return out
