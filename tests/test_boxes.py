"""Rotated IoU / NMS / points-in-boxes: C oracle vs an independent polygon clipper (CPU); HIP vs oracle (GPU)."""
import numpy as np
import pytest
import torch

from oracle import boxes as ob


def _rand_boxes(rng, n, spread=20.0):
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rng.uniform(-spread, spread, (n, 2))
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3] = rng.uniform(1.5, 5.0, n)
    b[:, 4] = rng.uniform(0.6, 2.5, n)
    b[:, 5] = rng.uniform(1.2, 2.0, n)
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    return b


def _corners(b):
    c = np.array([[-0.5, -0.5], [0.5, -0.5], [0.5, 0.5], [-0.5, 0.5]]) * b[3:5]
    ca, sa = np.cos(b[6]), np.sin(b[6])
    return c @ np.array([[ca, sa], [-sa, ca]]) + b[0:2]


def _clip_area(pa, pb):
    """Sutherland-Hodgman intersection area of two convex polygons (float64) — independent of the reference's method."""
    out = [tuple(p) for p in pa]
    for i in range(len(pb)):
        a, b = pb[i], pb[(i + 1) % len(pb)]
        inp, out = out, []
        if not inp:
            break
        def side(p):
            return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        for j in range(len(inp)):
            p, q = inp[j], inp[(j + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if sp * sq < 0:
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    if len(out) < 3:
        return 0.0
    x, y = np.array([p[0] for p in out]), np.array([p[1] for p in out])
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def test_oracle_overlap_vs_polygon_clipping_and_known_answer():
    rng = np.random.default_rng(0)
    a, b = _rand_boxes(rng, 40, 6.0), _rand_boxes(rng, 30, 6.0)
    ov = ob.boxes_overlap_bev(a, b)
    for i in range(len(a)):
        for j in range(len(b)):
            ref = _clip_area(_corners(a[i].astype(np.float64)), _corners(b[j].astype(np.float64)))
            # the reference counts corners within 1e-2 of the other box as inside: allow margin * perimeter
            assert abs(ov[i, j] - ref) <= 5e-4 + 0.02 * (a[i, 3] + a[i, 4] + b[j, 3] + b[j, 4]) * (ref < 0.5) + 1e-3 * ref, (i, j, ov[i, j], ref)
    # value the reference's own compiled iou3d_cpu.cpp returned in the survey session (0.4421)
    hand = np.array([[0, 0, 0, 4, 2, 1.5, 0], [1, 0.5, 0, 4, 2, 1.5, 0.3]], np.float32)
    iou = ob.boxes_iou_bev(hand, hand)
    assert abs(iou[0, 0] - 1) < 1e-6 and abs(iou[1, 0] - 0.4421) < 1e-4


def test_oracle_nms_and_points_in_boxes():
    rng = np.random.default_rng(1)
    b = _rand_boxes(rng, 200, 8.0)
    keep = ob.nms(b, 0.1)
    iou = ob.boxes_iou_bev(b[keep], b[keep])
    np.fill_diagonal(iou, 0)
    assert (iou <= 0.1 + 1e-6).all() and keep[0] == 0
    pts = rng.uniform(-10, 10, (2, 500, 3)).astype(np.float32)
    pts[:, :, 2] *= 0.2
    boxes = np.stack([_rand_boxes(rng, 12, 8.0), _rand_boxes(rng, 12, 8.0)])
    idx = ob.points_in_boxes(pts, boxes)
    assert idx.min() == -1 and idx.max() < 12 and (idx >= 0).sum() > 5


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_overlap_iou_iou3d(cuda, hip_lib):
    from seevcn_amd.pcdet.ops.iou3d_nms import iou3d_nms_utils as u
    rng = np.random.default_rng(2)
    a, b = _rand_boxes(rng, 300, 12.0), _rand_boxes(rng, 77, 12.0)
    b[:5] = a[:5]                                     # identical boxes
    b[5, :] = a[5, :]; b[5, 0] += a[5, 3]             # touching along x (axis-aligned after rotation)
    ta, tb = torch.from_numpy(a).to(cuda), torch.from_numpy(b).to(cuda)
    np.testing.assert_allclose(u.boxes_overlap_bev(ta, tb).cpu().numpy(), ob.boxes_overlap_bev(a, b), rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(u.boxes_iou_bev(ta, tb).cpu().numpy(), ob.boxes_iou_bev(a, b), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(u.boxes_iou3d_gpu(ta, tb).cpu().numpy(), ob.boxes_iou3d(a, b), rtol=1e-3, atol=1e-4)
    # the one-launch 3-D IoU == the reference's chain of torch ops around the overlap kernel, bit for bit; batched over scenes, rows of 8 floats
    fused = u.boxes_iou3d_gpu(ta, tb)
    saved, u.FUSED_IOU3D = u.FUSED_IOU3D, False
    try:
        assert torch.equal(fused, u.boxes_iou3d_gpu(ta, tb))
    finally:
        u.FUSED_IOU3D = saved
    b8 = torch.cat([tb, torch.full((len(b), 1), 3.0, device=cuda)], dim=1)                  # ground-truth layout: a class column behind the box
    both = u.boxes_iou3d_batch(torch.stack([ta, ta.flip(0)]), torch.stack([b8, b8.flip(0)]))
    assert torch.equal(both[0], fused) and torch.equal(both[1], fused.flip(0).flip(1))
    assert u.boxes_iou3d_gpu(ta[:0], tb).shape == (0, 77) and u.boxes_iou3d_batch(ta[None], b8[None, :0]).shape == (1, 300, 0)
    assert u.boxes_iou_bev(ta[:0], tb).shape == (0, 77)
    cpu = u.boxes_iou_bev_cpu(a, b)                                   # numpy in / numpy out, like the reference's CPU entry point
    assert isinstance(cpu, np.ndarray) and np.array_equal(cpu, u.boxes_iou_bev(ta, tb).cpu().numpy())
    assert isinstance(u.boxes_iou_bev_cpu(torch.from_numpy(a), torch.from_numpy(b)), torch.Tensor)
    with pytest.raises(AssertionError):
        u.boxes_iou_bev_cpu(ta, tb)


@pytest.mark.gpu
@pytest.mark.parametrize("n,thr", [(50, 0.1), (700, 0.7), (4096, 0.8), (9000, 0.8)])
def test_hip_nms_matches_oracle(cuda, hip_lib, n, thr):
    from seevcn_amd.pcdet.ops.iou3d_nms import iou3d_nms_utils as u
    rng = np.random.default_rng(n)
    b = _rand_boxes(rng, n, 30.0)
    b[: n // 3, 0:2] = b[n // 3: 2 * (n // 3), 0:2][: n // 3] + rng.normal(0, 0.3, (n // 3, 2))   # clusters of near-duplicates
    s = rng.uniform(size=n).astype(np.float32)
    order = np.argsort(-s, kind="stable")
    keep, _ = u.nms_gpu(torch.from_numpy(b).to(cuda), torch.from_numpy(s).to(cuda), thr)
    ref = order[ob.nms(b[order], thr)]
    got = keep.cpu().numpy()
    # scores are distinct with probability 1, so the sort order is unambiguous
    assert np.array_equal(got, ref)
    keep_n, _ = u.nms_normal_gpu(torch.from_numpy(b).to(cuda), torch.from_numpy(s).to(cuda), thr)
    assert np.array_equal(keep_n.cpu().numpy(), order[ob.nms(b[order], thr, normal=True)])
    kp, _ = u.nms_gpu(torch.from_numpy(b).to(cuda), torch.from_numpy(s).to(cuda), thr, pre_maxsize=min(n, 512))
    assert np.array_equal(kp.cpu().numpy(), order[:512][ob.nms(b[order[:512]], thr)])
    # max_keep: the sweep stops early and returns exactly the prefix (what NMS_POST_MAXSIZE callers slice off anyway)
    for mk in (1, 37, 64, 65, len(ref), len(ref) + 10):
        km, _ = u.nms_gpu(torch.from_numpy(b).to(cuda), torch.from_numpy(s).to(cuda), thr, max_keep=mk)
        assert np.array_equal(km.cpu().numpy(), ref[:mk]), mk


@pytest.mark.gpu
def test_hip_points_in_boxes(cuda, hip_lib):
    from seevcn_amd.pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils as r
    rng = np.random.default_rng(3)
    pts = rng.uniform(-12, 12, (3, 5000, 3)).astype(np.float32)
    pts[:, :, 2] *= 0.15
    boxes = np.stack([_rand_boxes(rng, 40, 10.0) for _ in range(3)])
    boxes[2, 20:] = 0                                   # zero-padded boxes (gt_boxes padding)
    out = r.points_in_boxes_gpu(torch.from_numpy(pts).to(cuda), torch.from_numpy(boxes).to(cuda)).cpu().numpy()
    assert np.array_equal(out, ob.points_in_boxes(pts, boxes))
