"""Deterministic inputs shared by make_pvrcnn_golden.py (reference side) and tests/test_pvrcnn.py (this repo's modules)."""
import numpy as np

SMALL = dict(num_keypoints=256, features_source=('bev', 'x_conv3', 'x_conv4', 'raw_points'), roi_per_image=32, nms_post_train=64,
             nms_pre_train=512, dp_ratio=0.0)


def make_inputs():
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet import model_cfgs as C
    rng = np.random.default_rng(21)
    pts, gt = synth.make_scene_batch(2, seed=2000, n_az=60)
    gt = gt[:, :10].copy()
    inten = rng.uniform(size=(len(pts), 1)).astype(np.float32)
    points = np.concatenate([pts, inten], 1)                                  # [b, x, y, z, intensity]
    out = {'points': points, 'gt_boxes': gt}
    out['spatial_features'] = rng.normal(size=(2, 32, 200, 176)).astype(np.float32)
    vs, rg = np.array([0.05, 0.05, 0.1]), np.array(C.KITTI_RANGE[:3])
    for name, factor, ch in (('x_conv3', 4, 64), ('x_conv4', 8, 64)):
        c = np.floor((pts[:, 1:4] - rg) / (vs * factor)).astype(np.int32)
        ok = (c >= 0).all(1) & (c[:, 0] < 1408 // factor) & (c[:, 1] < 1600 // factor) & (c[:, 2] < 41 // factor + 1)
        idx = np.unique(np.concatenate([pts[ok, 0:1].astype(np.int32), c[ok][:, [2, 1, 0]]], 1), axis=0)
        out[name + '_indices'] = idx.astype(np.int32)
        out[name + '_features'] = rng.normal(size=(len(idx), ch)).astype(np.float32)
    # first-stage proposals: jittered copies of the ground truth + random boxes
    boxes, scores = [], []
    for b in range(2):
        g = gt[b][gt[b, :, 3] > 0][:, :7]
        rep = np.repeat(g, 12, axis=0) + rng.normal(0, 1, (len(g) * 12, 7)).astype(np.float32) * np.array([0.25, 0.25, 0.1, 0.1, 0.05, 0.05, 0.1], np.float32)
        rnd = np.concatenate([rng.uniform([0, -40, -2], [70, 40, 0], (200, 3)), rng.uniform([1.5, 0.6, 1.2], [4.5, 2, 2], (200, 3)),
                              rng.uniform(-3, 3, (200, 1))], 1).astype(np.float32)
        bx = np.concatenate([rep, rnd])[:400]
        boxes.append(np.concatenate([bx, np.zeros((400 - len(bx), 7), np.float32)]))
        scores.append(rng.normal(size=(400, 3)).astype(np.float32))
    out['batch_box_preds'] = np.stack(boxes).astype(np.float32)
    out['batch_cls_preds'] = np.stack(scores).astype(np.float32)
    return out
