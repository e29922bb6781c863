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
