"""Point isolation: image projection + FOV, mask lookup, range-adaptive DBSCAN, oriented-box crops, multi-camera
merge.  Pinned to the reference's own code through tests/golden/isolation.npz (make_isolation_golden.py); the open3d calls inside
are parity-unpinned (served by the oracle when the fixture was made) and cross-checked here against sklearn / scipy."""
import os

import numpy as np
import pytest
import torch

from isolation_inputs import (CLASSES, CUSTOM_IMG_SHAPE, IMG_SHAPE, MIN_LIDAR_PTS, PC_ISOLATION, custom_calibration, make_inputs,
                              multi_camera_instances)
from oracle import isolation as oiso
from oracle.postprocess import dbscan_labels


def _split(g, prefix):
    rows, counts = g[prefix + "_rows"], g[prefix + "_counts"]
    return np.split(rows, np.cumsum(counts)[:-1]) if len(counts) else []


@pytest.fixture(scope="module")
def inputs():
    return make_inputs()


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "isolation.npz"))


def _oracle_imgfov(inp):
    c = inp['calib']
    fov, pts_img, rect = oiso.project_velo_to_image_kitti(inp['points'], c['Tr_velo2cam'], c['R0'], c['P2'], IMG_SHAPE[0], IMG_SHAPE[1], 1.0)
    return fov, pts_img, rect


def test_oracle_projection_matches_reference_golden(inputs, golden):
    fov, pts_img, rect = _oracle_imgfov(inputs)
    assert np.array_equal(fov, golden["fov_inds"]) and 0 < fov.sum() < len(fov)
    assert np.array_equal(pts_img, golden["pts_img"])                       # floor(u,v): bit-exact pixels
    np.testing.assert_allclose(rect, golden["pc_cam"], rtol=0, atol=1e-12)  # BLAS summation order


@pytest.mark.parametrize("model", ["pinhole", "equidistant"])
def test_oracle_custom_camera_projection_matches_reference_golden(inputs, golden, model):
    c = custom_calibration(model)
    fov, pts_img, uv = oiso.project_custom_camera(inputs['points'], c['intrinsic'], c['extrinsic'], c['distcoeff'], *CUSTOM_IMG_SHAPE, camera_model=model)
    assert np.array_equal(fov, golden[f"custom_{model}_fov_inds"]) and 0 < (~fov).sum() < len(fov)
    assert np.array_equal(pts_img, golden[f"custom_{model}_pts_img"])              # rounded pixels and depth: bit-exact
    np.testing.assert_allclose(uv, golden[f"custom_{model}_pc_cam"], rtol=0, atol=1e-9)


def test_oracle_mask_lookup_matches_reference_golden(inputs, golden):
    fov, pts_img, _ = _oracle_imgfov(inputs)
    lidar = inputs['points'][fov]
    kept = [i for i in inputs['instances'] if i['segmentation']]
    for mode in ("mask", "bbox"):
        if mode == "mask":
            lists = oiso.pts_in_masks(pts_img, masks=[i['bin_mask'] for i in kept])
        else:
            rects = []
            for i in kept:
                b = np.array(i['bbox'])
                b[2:4] = b[0:2] + b[2:4]
                rects.append([max(int(b[0]), 0), max(int(b[1]), 0), min(int(b[2]), IMG_SHAPE[1]), min(int(b[3]), IMG_SHAPE[0])])
            lists = oiso.pts_in_masks(pts_img, rects=rects)
        got = [(lidar[l], pts_img[l], i['box_id']) for l, i in zip(lists, kept) if len(l)]
        want_l, want_uv = _split(golden, f"inmask_{mode}_lidar"), _split(golden, f"inmask_{mode}_uv")
        assert [g[2] for g in got] == list(golden[f"inmask_{mode}_box_id"]) and len(got) >= 5
        for (l, uv, _), wl, wuv in zip(got, want_l, want_uv):
            assert np.array_equal(l, wl) and np.array_equal(uv, wuv)


def test_oracle_isolate_det_and_gt_match_reference_golden(inputs, golden):
    clouds = _split(golden, "inmask_mask_lidar")
    got = oiso.isolate_det_pts(clouds, PC_ISOLATION['VRES'], PC_ISOLATION['EPS_SCALING'], PC_ISOLATION['MIN_EPS'], PC_ISOLATION['MAX_EPS'], 10)
    want = _split(golden, "det_instances")
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    assert sum(len(w) for w in want) < sum(len(c) for c in clouds)          # the clustering removed background points
    boxes = [b for b, n in zip(inputs['sample_infos']['annos']['gt_boxes_lidar'], inputs['sample_infos']['annos']['name']) if n in CLASSES]
    crops, labels = oiso.isolate_gt_pts(inputs['points'], boxes, MIN_LIDAR_PTS)
    want = _split(golden, "gt_crops")
    assert len(crops) == len(want) >= 3 and all(np.array_equal(a, b) for a, b in zip(crops, want))
    assert np.array_equal(np.stack(labels), golden["gt_labels"])
    merged = oiso.merge_multi_camera_detections(multi_camera_instances())
    want = _split(golden, "merged")
    assert len(merged) == len(want) == 4 and all(np.array_equal(a, b) for a, b in zip(merged, want))


def _cluster_case(rng, n, spread=0.25):
    centres = rng.uniform(-6, 6, (5, 3))
    sizes = rng.multinomial(n - n // 10, [0.45, 0.25, 0.15, 0.1, 0.05])
    pts = [rng.normal(0, spread, (s, 3)) + c for s, c in zip(sizes, centres)]
    pts.append(rng.uniform(-8, 8, (n - sum(sizes), 3)))
    return rng.permutation(np.vstack(pts)).astype(np.float32)


def test_oracle_dbscan_min_points_3_vs_sklearn():
    from sklearn.cluster import DBSCAN
    rng = np.random.default_rng(3)
    for n, eps, mp in [(300, 0.3, 3), (700, 0.22, 5), (150, 0.5, 3)]:
        x = _cluster_case(rng, n)
        mine = dbscan_labels(x, eps, mp)
        sk = DBSCAN(eps=eps, min_samples=mp).fit(x.astype(np.float64))
        core = np.zeros(n, bool)
        core[sk.core_sample_indices_] = True
        assert np.array_equal(mine >= 0, sk.labels_ >= 0)
        # same partition of the core points (labels are numbered differently; border ties may go to different clusters)
        pairs = set(zip(mine[core].tolist(), sk.labels_[core].tolist()))
        assert len(pairs) == len(set(mine[core].tolist())) == len(set(sk.labels_[core].tolist()))


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_projection_and_masks_bitexact_vs_reference_golden(inputs, golden, cuda, hip_lib):
    from seevcn_amd.vcn import isolation as I
    calib = I.Calibration(inputs['calib'])
    imgfov = I.map_pointcloud_to_image(inputs['points'], calib, IMG_SHAPE, min_dist=1.0)
    assert np.array_equal(imgfov["fov_inds"], golden["fov_inds"])
    assert np.array_equal(imgfov["pts_img"], golden["pts_img"])
    assert np.array_equal(imgfov["pc_lidar"], inputs['points'][golden["fov_inds"]])
    np.testing.assert_allclose(imgfov["pc_cam"], golden["pc_cam"], rtol=1e-6, atol=1e-5)     # float32 storage of the rectified coordinates
    for mode, use_bbox in (("mask", False), ("bbox", True)):
        proj = I.get_pts_in_mask(None, inputs['instances'], imgfov, shrink_percentage=0, use_bbox=use_bbox)
        want_l, want_uv = _split(golden, f"inmask_{mode}_lidar"), _split(golden, f"inmask_{mode}_uv")
        assert [l['box_id'] for l in proj["img_labels"]] == list(golden[f"inmask_{mode}_box_id"])
        assert len(proj["lidar_xyz"]) == len(want_l) == len(proj["cam_xyz"])
        for l, uv, wl, wuv in zip(proj["lidar_xyz"], proj["img_uv"], want_l, want_uv):
            assert np.array_equal(l, wl) and np.array_equal(uv, wuv)


@pytest.mark.gpu
@pytest.mark.parametrize("index_form", ["mask", "ascending", "shuffled"])
def test_hip_waymo_precomputed_projection_feeds_the_same_mask_lookup(inputs, golden, cuda, hip_lib, tmp_path, index_form):
    """The Waymo branch (waymo_objects.py:170-186): the projection is LOADED (image_pc + fov_inds .npy per frame and camera), then the same
    get_pts_in_mask as every other dataset runs on it.  Fed with the reference's own projection of the golden frame -- as a boolean mask, as an
    ascending index array and as a shuffled one, through the directory layout the reference reads -- the per-instance point lists must equal the
    reference's (the shuffled form in its own FOV-row order)."""
    from seevcn_amd.vcn import isolation as I
    pts, fov, pix = inputs['points'], golden["fov_inds"], golden["pts_img"]
    if index_form == "mask":
        fi, px = fov, pix
    else:
        fi = np.nonzero(fov)[0]
        px = pix
        if index_form == "shuffled":
            perm = np.random.default_rng(0).permutation(len(fi))
            fi, px = fi[perm], pix[perm]
    for sub, arr in (("image_pc", px), ("fov_inds", fi)):
        d = tmp_path / "image_lidar_projections" / sub / "FRONT"
        d.mkdir(parents=True)
        np.save(d / "segment-1_0007.npy", arr)
    imgfov = I.waymo_map_pointcloud_to_image(tmp_path, "segment-1", 7, "FRONT", pts, img_shape=IMG_SHAPE)
    assert imgfov["pc_cam"] is None and np.array_equal(imgfov["pc_lidar"], pts[fi]) and np.array_equal(imgfov["pts_img"], px)
    for mode, use_bbox in (("mask", False), ("bbox", True)):
        proj = I.get_pts_in_mask(None, inputs['instances'], imgfov, shrink_percentage=0, use_bbox=use_bbox)
        want_l, want_uv = _split(golden, f"inmask_{mode}_lidar"), _split(golden, f"inmask_{mode}_uv")
        assert [l['box_id'] for l in proj["img_labels"]] == list(golden[f"inmask_{mode}_box_id"]) and proj["cam_xyz"] == []
        for l, uv, wl, wuv in zip(proj["lidar_xyz"], proj["img_uv"], want_l, want_uv):
            if index_form == "shuffled":                           # same point set per instance, in the shuffled arrays' row order
                key = lambda a: a[np.lexsort(a.T[::-1])]
                assert np.array_equal(key(l), key(wl)) and np.array_equal(key(np.asarray(uv)), key(np.asarray(wuv)))
            else:
                assert np.array_equal(l, wl) and np.array_equal(uv, wuv)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["pinhole", "equidistant"])
def test_hip_custom_camera_projection_vs_reference_golden(inputs, golden, cuda, hip_lib, model):
    from seevcn_amd.vcn import isolation as I
    imgfov = I.map_pointcloud_to_image_custom(inputs['points'], custom_calibration(model), CUSTOM_IMG_SHAPE, camera_model=model)
    assert np.array_equal(imgfov["fov_inds"], golden[f"custom_{model}_fov_inds"])
    assert np.array_equal(imgfov["pts_img"], golden[f"custom_{model}_pts_img"])
    assert np.array_equal(imgfov["pc_lidar"], inputs['points'][golden[f"custom_{model}_fov_inds"]])
    np.testing.assert_allclose(imgfov["pc_cam"], golden[f"custom_{model}_pc_cam"], rtol=0, atol=1e-9)
    with pytest.raises(NotImplementedError):
        I.map_pointcloud_to_image_custom(inputs['points'], custom_calibration(model), CUSTOM_IMG_SHAPE, camera_model="spherical")


@pytest.mark.gpu
def test_hip_isolate_det_gt_merge_bitexact_vs_reference_golden(inputs, golden, cuda, hip_lib):
    from seevcn_amd.vcn import isolation as I
    proj = {"lidar_xyz": _split(golden, "inmask_mask_lidar"), "img_uv": _split(golden, "inmask_mask_uv")}
    got = I.isolate_det_pts([proj], PC_ISOLATION['VRES'], PC_ISOLATION['EPS_SCALING'], PC_ISOLATION['MIN_EPS'], PC_ISOLATION['MAX_EPS'], min_cluster=10)
    want = _split(golden, "det_instances")
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    pcd_gtboxes = I.populate_gtboxes(inputs['sample_infos'], 'kitti', CLASSES, add_ground_lift=True, ground_lift_height=0.1)
    pcd_gtboxes['pcd'] = inputs['points']
    crops, labels = I.isolate_gt_pts(pcd_gtboxes, MIN_LIDAR_PTS)
    want = _split(golden, "gt_crops")
    assert len(crops) == len(want) and all(a.dtype == np.float64 and np.array_equal(a, b) for a, b in zip(crops, want))
    assert np.array_equal(np.stack(labels), golden["gt_labels"])
    merged = I.merge_multi_camera_detections(multi_camera_instances())
    want = _split(golden, "merged")
    assert len(merged) == len(want) and all(np.array_equal(a, b) for a, b in zip(merged, want))


@pytest.mark.gpu
def test_hip_largest_cluster_vs_oracle_edge_cases(cuda, hip_lib):
    """Instances of every size class in ONE launch: <= min_cluster points, all noise, LDS path, > 4096 points (scratch path),
    min_points 1..5, fixed and range-adaptive eps; indices bit-exact against the oracle."""
    from seevcn_amd.vcn import isolation as I
    rng = np.random.default_rng(11)
    clouds = [_cluster_case(rng, n) + np.float32(off) for n, off in [(400, 10), (1500, 25), (9, 5), (64, 40), (4500, 15), (4096, 30)]]
    clouds.append((np.arange(60, dtype=np.float32).reshape(20, 3) * 7 + 3))                  # all noise
    counts = np.array([len(c) for c in clouds], np.int32)
    starts = np.concatenate([[0], np.cumsum(counts[:-1])]).astype(np.int64)
    pts = torch.from_numpy(np.vstack(clouds)).to(cuda)
    for mp, fixed in [(3, None), (1, 0.2), (2, 0.3), (5, 0.35)]:
        kw = dict(fixed_eps=fixed) if fixed is not None else dict(vres=0.4, eps_scaling=4, min_eps=0.0, max_eps=1.0)
        out_local, out_count, out_eps = I.largest_clusters_device(pts, torch.from_numpy(starts).to(cuda), torch.from_numpy(counts).to(cuda),
                                                                  int(counts.max()), min_points=mp, min_cluster=10, **kw)
        out_local, out_count, out_eps = out_local.cpu().numpy(), out_count.cpu().numpy(), out_eps.cpu().numpy()
        for g, c in enumerate(clouds):
            if len(c) <= 10:
                assert out_count[g] == 0
                continue
            eps = fixed if fixed is not None else oiso.instance_eps(c, 0.4, 4, 0.0, 1.0)
            assert out_eps[g] == eps
            idx = oiso.largest_cluster_indices(c, eps, mp)
            if idx is None:
                assert out_count[g] == 0
            else:
                assert out_count[g] == len(idx) and np.array_equal(out_local[starts[g]:starts[g] + len(idx)], idx), (g, mp)
    # point_index indirection + the caller's bound
    perm = rng.permutation(len(pts)).astype(np.int32)
    inv = np.argsort(perm).astype(np.int32)
    out_local2, out_count2, _ = I.largest_clusters_device(pts[torch.from_numpy(perm).to(cuda)].contiguous(), torch.from_numpy(starts).to(cuda),
                                                          torch.from_numpy(counts).to(cuda), int(counts.max()), point_index=torch.from_numpy(inv).to(cuda),
                                                          fixed_eps=0.35, min_points=5, min_cluster=10)
    assert np.array_equal(out_count2.cpu().numpy(), out_count)
    _, bad, _ = I.largest_clusters_device(pts, torch.from_numpy(starts).to(cuda), torch.from_numpy(counts).to(cuda), 1000, fixed_eps=0.3)
    assert list(bad.cpu().numpy() == -1) == [bool(c > 1000) for c in counts]


@pytest.mark.gpu
def test_hip_crop_boxes_vs_oracle_random(cuda, hip_lib):
    from seevcn_amd.vcn import isolation as I
    rng = np.random.default_rng(5)
    pts = rng.uniform(-20, 20, (30000, 4)).astype(np.float32)
    boxes = [I.OrientedBox(rng.uniform(-15, 15, 3), I.gtbox_to_corners([0, 0, 0, 1, 1, 1, rng.uniform(-3, 3)])[1], rng.uniform(0.5, 12, 3))
             for _ in range(9)]
    boxes.append(I.OrientedBox([1, -2, 0], I.gtbox_to_corners([0, 0, 0, 1, 1, 1, 0.7])[1], [30, 25, 38]))   # > 1024 points: several select rounds
    boxes.append(I.OrientedBox([100, 100, 100], np.eye(3), [1, 1, 1]))                     # empty crop
    dev = torch.from_numpy(pts).to(cuda)
    index, count = I.crop_boxes_device(dev, boxes)                                          # row stride 4
    small_index, small_count = I.crop_boxes_device(dev, boxes, cap=16)
    index, count = index.cpu().numpy(), count.cpu().numpy()
    assert np.array_equal(small_count.cpu().numpy(), count)
    for g, b in enumerate(boxes):
        want = oiso.crop_oriented_box(pts, b.center, b.R, b.extent)
        assert count[g] == len(want) and np.array_equal(index[g, :count[g]], want)
        assert np.array_equal(small_index.cpu().numpy()[g, :min(16, len(want))], want[:16])
    assert count[-1] == 0 and count[:-1].max() > 1024


def test_isolation_refuses_cpu(hip_lib):
    from seevcn_amd.vcn import isolation as I
    with pytest.raises((RuntimeError, AssertionError, ValueError)):
        I.crop_boxes_device(torch.zeros((10, 3)), [I.OrientedBox([0, 0, 0], np.eye(3), [1, 1, 1])])


# ------------------------------------------------------------------------------------------ polygons -> masks (annToMask)
def _random_polygons(rng, h, w, n):
    """star-shaped and arbitrary polygons with fractional vertices, some leaving the image"""
    out = []
    for i in range(n):
        k = int(rng.integers(3, 40))
        if i % 2:
            c = rng.uniform([0, 0], [w, h])
            ang = np.sort(rng.uniform(0, 2 * np.pi, k))
            r = rng.uniform(2, 0.6 * min(h, w), k)
            pts = np.stack([c[0] + r * np.cos(ang), c[1] + r * np.sin(ang)], 1)
        else:
            pts = rng.uniform([-5, -5], [w + 5, h + 5], (k, 2))
        if i % 3 == 0:
            pts = np.round(pts)                                                  # integer vertices: boundaries exactly on pixel corners
        out.append([float(t) for t in pts.reshape(-1)])
    return out


def test_oracle_polygon_mask_known_answers():
    """oracle/coco_mask.py (a restatement of pycocotools' rleFrPoly / decode, PARITY UNPINNED: the library is absent) against answers derived by hand
    from the algorithm: an integer rectangle (x0, y0)-(x1, y1) covers the pixels [x0, x1) x [y0, y1); a polygon along the image border covers the
    whole image (its runs end on the border, equal run boundaries cancel); an empty / degenerate polygon covers nothing."""
    from oracle import coco_mask as cm
    m = cm.ann_to_mask({"segmentation": [[1, 1, 4, 1, 4, 3, 1, 3]]}, 5, 6)
    want = np.zeros((5, 6), np.uint8)
    want[1:3, 1:4] = 1
    assert np.array_equal(m, want)
    assert cm.rle_from_polygon([1, 1, 4, 1, 4, 3, 1, 3], 5, 6) == [6, 2, 3, 2, 3, 2, 12]      # runs worked out by hand in the test's history (column-major)
    assert np.array_equal(cm.ann_to_mask({"segmentation": [[0, 0, 6, 0, 6, 5, 0, 5]]}, 5, 6), np.ones((5, 6), np.uint8))
    assert cm.ann_to_mask({"segmentation": [[2, 2, 2, 2, 2, 2]]}, 5, 6).sum() == 0
    # two parts of one annotation: their union; an uncompressed RLE dict: its runs, column-major
    two = cm.ann_to_mask({"segmentation": [[0, 0, 2, 0, 2, 2, 0, 2], [3, 3, 6, 3, 6, 5, 3, 5]]}, 5, 6)
    want = np.zeros((5, 6), np.uint8)
    want[0:2, 0:2] = 1
    want[3:5, 3:6] = 1
    assert np.array_equal(two, want)
    assert np.array_equal(cm.ann_to_mask({"segmentation": {"size": [5, 6], "counts": [6, 2, 3, 2, 3, 2, 12]}}, 5, 6), cm.ann_to_mask({"segmentation": [[1, 1, 4, 1, 4, 3, 1, 3]]}, 5, 6))


@pytest.mark.gpu
def test_hip_polygon_masks_bitexact_vs_oracle(cuda, hip_lib):
    """sv_polygons_to_masks (boundary walk + parity scan) == the oracle's run-length route, pixel for pixel: the hand-derived cases, 60 random polygons
    (integer and fractional vertices, concave, self-intersecting, partly outside) on a 93 x 131 image, multi-part instances, and a KITTI-sized image."""
    from oracle import coco_mask as cm
    from seevcn_amd.vcn import isolation as I
    inst = [{"segmentation": [[1, 1, 4, 1, 4, 3, 1, 3]]}, {"segmentation": [[0, 0, 6, 0, 6, 5, 0, 5]]}, {"segmentation": [[2, 2, 2, 2, 2, 2]]},
            {"segmentation": [[0, 0, 2, 0, 2, 2, 0, 2], [3, 3, 6, 3, 6, 5, 3, 5]]}, {"segmentation": {"size": [5, 6], "counts": [6, 2, 3, 2, 3, 2, 12]}}]
    got = I.instance_masks_device(inst, 5, 6, cuda).cpu().numpy()
    for g, a in zip(got, inst):
        assert np.array_equal(g, cm.ann_to_mask(a, 5, 6)), a
    rng = np.random.default_rng(11)
    h, w = 93, 131
    polys = _random_polygons(rng, h, w, 60)
    inst = [{"segmentation": [p]} for p in polys[:40]] + [{"segmentation": polys[40 + 4 * i:44 + 4 * i]} for i in range(5)]
    got = I.instance_masks_device(inst, h, w, cuda).cpu().numpy()
    for i, (g, a) in enumerate(zip(got, inst)):
        want = cm.ann_to_mask(a, h, w)
        assert np.array_equal(g, want), (i, int((g != want).sum()))
    assert 0 < got.mean() < 1
    h, w = 375, 1242
    polys = _random_polygons(np.random.default_rng(12), h, w, 6)
    got = I.instance_masks_device([{"segmentation": [p]} for p in polys], h, w, cuda).cpu().numpy()
    for g, p in zip(got, polys):
        assert np.array_equal(g, cm.ann_to_mask({"segmentation": [p]}, h, w))


@pytest.mark.gpu
def test_hip_get_pts_in_mask_takes_coco_polygons(inputs, golden, cuda, hip_lib):
    """get_pts_in_mask with instances that carry ONLY polygons (as the reference's detections do before annToMask, shared_utils.py:62-69): the masks
    are rasterised on the device, the selected points equal those of the same masks handed in as 'bin_mask' (made by the oracle), and every label
    comes back with its binary mask like the reference's."""
    from oracle import coco_mask as cm
    from seevcn_amd.vcn import isolation as I
    calib = I.Calibration(inputs['calib'])
    imgfov = I.map_pointcloud_to_image(inputs['points'], calib, IMG_SHAPE, min_dist=1.0)
    rng = np.random.default_rng(3)
    polys = []
    for inst in inputs['instances'][:6]:
        x0, y0, bw, bh = inst['bbox']
        k = 12
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        pts = np.stack([x0 + bw / 2 + 0.7 * bw * np.cos(ang), y0 + bh / 2 + 0.7 * bh * np.sin(ang)], 1)
        polys.append([float(t) for t in pts.reshape(-1)])
    only_poly = [{"segmentation": [p], "bbox": i["bbox"], "category_id": 1, "box_id": i["box_id"]} for p, i in zip(polys, inputs['instances'])]
    with_mask = [dict(a, bin_mask=cm.ann_to_mask(a, *IMG_SHAPE)) for a in only_poly]
    a = I.get_pts_in_mask(None, only_poly, imgfov)
    b = I.get_pts_in_mask(None, with_mask, imgfov)
    assert len(a["lidar_xyz"]) == len(b["lidar_xyz"]) > 0
    for la, lb, ia, ib in zip(a["lidar_xyz"], b["lidar_xyz"], a["img_labels"], b["img_labels"]):
        assert np.array_equal(la, lb) and np.array_equal(ia["bin_mask"], ib["bin_mask"])


def test_oracle_shrunken_mask_known_answers():
    """oracle/coco_mask.ann_to_mask_shrunk (the region of shapely's buffer(-d) sampled at the pixel centres -- PARITY UNPINNED and known to differ from
    GEOS's rasterised vertex list in a band of about one pixel): an integer rectangle shrinks to the pixels at least d from its four edges; the
    distance follows shared_utils.py:295-306 (percentage of the bounding box's half diagonal); a part that shrinks to nothing leaves the instance
    unshrunken (shared_utils.py:325-326); percentage 0 and RLE dicts are untouched."""
    from oracle import coco_mask as cm
    rect = [10, 20, 50, 20, 50, 50, 10, 50]
    assert abs(cm.shrink_distance(rect, 10) - 0.5 * np.hypot(40, 30) * 0.1) < 1e-12           # 2.5
    m = cm.ann_to_mask_shrunk({"segmentation": [rect]}, 64, 64, 10)
    want = np.zeros((64, 64), np.uint8)
    want[23:48, 13:48] = 1                                     # pixel centres x in [12.5, 47.5], y in [22.5, 47.5]
    assert np.array_equal(m, want)
    assert np.array_equal(cm.ann_to_mask_shrunk({"segmentation": [rect]}, 64, 64, 0), cm.ann_to_mask({"segmentation": [rect]}, 64, 64))
    sliver = [5, 5, 60, 5, 60, 7, 5, 7]                        # half diagonal 27.5: 10 % = 2.75 > its half height
    two = {"segmentation": [rect, sliver]}
    assert np.array_equal(cm.ann_to_mask_shrunk(two, 64, 64, 10), cm.ann_to_mask(two, 64, 64))
    # an L shape: the reflex corner's circle of radius d is cut out
    ell = [0, 0, 40, 0, 40, 20, 20, 20, 20, 40, 0, 40]
    m = cm.ann_to_mask_shrunk({"segmentation": [ell]}, 48, 48, 15)
    d = cm.shrink_distance(ell, 15)
    assert abs(d - 0.5 * np.hypot(40, 40) * 0.15) < 1e-12                                  # 4.24
    assert m[15, 15] == 1 and m[10, 30] == 1 and m[30, 10] == 1                             # >= d from every edge
    assert m[18, 18] == 0                                                                   # 2.83 from the reflex corner (20, 20): inside its circle
    assert m[10, 37] == 0 and m[3, 10] == 0                                                 # 3 from the edge x = 40 / y = 0
    assert m.sum() < cm.ann_to_mask({"segmentation": [ell]}, 48, 48).sum()


@pytest.mark.gpu
def test_hip_masks_within_distance_equal_the_oracle(cuda, hip_lib):
    """sv_polygons_to_masks_shrunk (the REGION form of the shrunken masks, round 5; get_pts_in_mask now takes the vertex-list form below) ==
    oracle/coco_mask.ann_to_mask_shrunk pixel for pixel: the hand cases (rectangle, L shape, a part that shrinks away -> unshrunken instance), 40 random
    polygons and multi-part instances at 3 % and 10 %."""
    from oracle import coco_mask as cm
    from seevcn_amd.vcn import isolation as I
    rect, sliver, ell = [10, 20, 50, 20, 50, 50, 10, 50], [5, 5, 60, 5, 60, 7, 5, 7], [0, 0, 40, 0, 40, 20, 20, 20, 20, 40, 0, 40]
    inst = [{"segmentation": [rect]}, {"segmentation": [ell]}, {"segmentation": [rect, sliver]}, {"segmentation": [sliver]}]
    for pct in (10, 3):
        got = I.instance_masks_within_distance(inst, 64, 64, cuda, pct).cpu().numpy()
        for g, a in zip(got, inst):
            assert np.array_equal(g, cm.ann_to_mask_shrunk(a, 64, 64, pct)), (pct, a)
    rng = np.random.default_rng(21)
    h, w = 93, 131
    polys = _random_polygons(rng, h, w, 52)
    inst = [{"segmentation": [p]} for p in polys[:40]] + [{"segmentation": polys[40 + 4 * i:44 + 4 * i]} for i in range(3)]
    for pct in (3, 10):
        got = I.instance_masks_within_distance(inst, h, w, cuda, pct).cpu().numpy()
        plain = I.instance_masks_device(inst, h, w, cuda).cpu().numpy()
        for i, (g, a) in enumerate(zip(got, inst)):
            want = cm.ann_to_mask_shrunk(a, h, w, pct)
            assert np.array_equal(g, want), (pct, i, int((g != want).sum()))
        assert got.sum() < plain.sum() and ((got == 1) <= (plain == 1)).all()


def _canonical_rings(seg_list):
    """int polygon lists -> sorted tuple of rings, each without its closing vertex and rotated to start at its smallest vertex (shapely's start vertex is
    its own business; the rasterisation does not depend on it)"""
    out = []
    for l in seg_list:
        pts = list(zip(l[0::2], l[1::2]))
        if len(pts) > 1 and pts[0] == pts[-1]:
            pts = pts[:-1]
        k = pts.index(min(pts))
        out.append(tuple(pts[k:] + pts[:k]))
    return tuple(sorted(out))


def test_oracle_polygon_buffer_known_answers():
    """oracle/polygon_buffer (shapely's Polygon.buffer(-d) + exterior + int() as shrink_instance_masks uses them, shared_utils.py:295-330; PARITY
    UNPINNED -- shapely / GEOS are not available) against answers derived by hand: a rectangle moves every edge inwards by d and truncates; an L
    keeps its five convex corners and turns the reflex one into 16 chords of the circle of radius d around it (quad_segs 16: one chord per pi / 32);
    a sliver thinner than 2 d is the empty polygon -> the function returns the list it was given (shared_utils.py:325-326), whatever the other
    parts were; a dumb-bell whose neck is thinner than 2 d splits into a MultiPolygon -> two lists."""
    from oracle import polygon_buffer as opb
    rect = [10, 20, 50, 20, 50, 50, 10, 50]
    assert abs(opb.shrink_distance(rect[0::2], rect[1::2], 10) - 2.5) < 1e-12
    assert _canonical_rings(opb.shrink_instance_masks([rect], 10)) == (((12, 22), (47, 22), (47, 47), (12, 47)),)
    got = opb.shrink_instance_masks([rect], 10)[0]
    assert got[:2] == got[-2:] and len(got) == 10                                      # exterior.coords repeats the first vertex
    ell = [0, 0, 40, 0, 40, 20, 20, 20, 20, 30, 0, 30]                                 # bounding box 40 x 30: half diagonal 25, 10 % = 2.5
    arc = [(int(20 + 2.5 * np.cos(-np.pi / 2 - s * np.pi / 32)), int(20 + 2.5 * np.sin(-np.pi / 2 - s * np.pi / 32))) for s in range(1, 16)]
    want = [(2, 2), (37, 2), (37, 17), (20, 17)] + arc + [(17, 20), (17, 27), (2, 27)]
    assert _canonical_rings(opb.shrink_instance_masks([ell], 10)) == (tuple(want),)
    sliver = [5, 5, 60, 5, 60, 7, 5, 7]                                                # half diagonal 27.5: 10 % = 2.75 > its half height
    parts = [rect, sliver]
    assert opb.shrink_instance_masks(parts, 10) is parts and opb.shrink_instance_masks([sliver], 10)[0] is sliver
    bell = [0, 0, 40, 0, 40, 18, 60, 18, 60, 0, 100, 0, 100, 40, 60, 40, 60, 22, 40, 22, 40, 40, 0, 40]      # neck 4 high; 6 % of 53.85 = 3.23
    two = opb.shrink_instance_masks([bell], 6)
    assert len(two) == 2
    boxes = sorted((min(l[0::2]), min(l[1::2]), max(l[0::2]), max(l[1::2])) for l in two)
    assert boxes == [(3, 3, 37, 36), (62, 3, 96, 36)]                                  # the lobes, 3.23 in from their sides; the arcs reach 3.23 into the neck
    assert opb.shrink_instance_masks([rect], 0) == [rect + rect[:2]]                   # buffer(0) of a valid ring: itself, closed


def test_polygon_buffer_equals_the_oracle_and_covers_the_eroded_region():
    """vcn/polygon_buffer.shrink_instance_masks (numpy, vectorised) == oracle/polygon_buffer (scalar restatement) ring for ring on random simple polygons
    with real and with integer vertices, 3 ... 20 %; and the region property that pins the construction without GEOS: every pixel centre inside the ring
    and more than d + 0.05 from it lies in the buffer's rings, no pixel centre nearer than d - 0.05 does (the arcs' chords sag 0.0012 d)."""
    from oracle import coco_mask as cm
    from oracle import polygon_buffer as opb
    from seevcn_amd.vcn import polygon_buffer as pb
    rng = np.random.default_rng(5)
    done = 0
    for trial in range(48):
        n = int(rng.integers(4, 48))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        rad = 40 + 25 * rng.random(n) * (rng.random() < 0.8) + 10 * np.sin(3 * ang)
        P = np.stack([80 + rad * np.cos(ang), 70 + rad * np.sin(ang)], 1)
        if trial % 2:
            P = np.round(P)
        pc = pb._clean_ring(P)
        if pc is None or len(pb._node(pc, 12)[0]) != len(pc):
            continue                                                                   # the rounded star crosses itself: not a polygon
        seg = [P.reshape(-1).tolist()]
        pct = float(rng.choice([3, 6, 10, 20]))
        a, b = pb.shrink_instance_masks(seg, pct), opb.shrink_instance_masks(seg, pct)
        assert (a is seg) == (b is seg)
        if a is not seg:
            assert _canonical_rings(a) == _canonical_rings(b), trial
        d = pb.shrink_distance(seg[0][0::2], seg[0][1::2], pct)
        assert abs(d - opb.shrink_distance(seg[0][0::2], seg[0][1::2], pct)) < 1e-12
        rings = pb.buffer_inward(P, d)
        yy, xx = np.mgrid[0:150, 0:170].astype(np.float64)
        px, py = xx.ravel(), yy.ravel()
        inside = pb._winding(px, py, pc, np.roll(pc, -1, axis=0)) != 0
        dist2 = cm._edge_distance2(px, py, seg[0])
        got = np.zeros(len(px), bool)
        for r in rings:
            got |= pb._winding(px, py, r[:-1], r[1:]) != 0
        assert not (inside & (dist2 >= (d + 0.05) ** 2) & ~got).any() and not ((~inside | (dist2 <= max(d - 0.05, 0.0) ** 2)) & got).any(), trial
        done += 1
    assert done >= 30


@pytest.mark.gpu
def test_hip_get_pts_in_mask_shrinks_like_the_reference(inputs, cuda, hip_lib):
    """get_pts_in_mask(shrink_percentage=3) (SHRINK_MASK_PERCENTAGE of every IMG_DET cfg; shared_utils.py:62-69): the instance's polygon list is replaced by
    the oracle's shrunken list (ring for ring), its mask is annToMask of THAT list pixel for pixel (oracle/coco_mask on oracle/polygon_buffer), the points
    are the points of that mask and fewer than without shrinking; append_mask_info counts the SHRUNKEN parts (a dumb-bell that splits counts 2); an instance
    with a part that vanishes keeps its original polygons and mask."""
    from oracle import coco_mask as cm
    from oracle import polygon_buffer as opb
    from seevcn_amd.vcn import isolation as I
    rect, sliver = [10, 20, 50, 20, 50, 50, 10, 50], [5, 5, 60, 5, 60, 7, 5, 7]
    bell = [0, 0, 40, 0, 40, 18, 60, 18, 60, 0, 100, 0, 100, 40, 60, 40, 60, 22, 40, 22, 40, 40, 0, 40]
    inst = [{"segmentation": [rect]}, {"segmentation": [bell]}, {"segmentation": [rect, sliver]}]
    for pct in (10, 6, 3):
        got = I.instance_masks_device(inst, 64, 110, cuda, pct).cpu().numpy()
        for g, a in zip(got, inst):
            want = cm.ann_to_mask({"segmentation": opb.shrink_instance_masks(a["segmentation"], pct)}, 64, 110)
            assert np.array_equal(g, want), (pct, a)
    rng = np.random.default_rng(21)
    h, w = 93, 131
    polys = _random_polygons(rng, h, w, 52)
    inst = [{"segmentation": [p]} for p in polys[:40]] + [{"segmentation": polys[40 + 4 * i:44 + 4 * i]} for i in range(3)]
    for pct in (3, 10):
        got = I.instance_masks_device(inst, h, w, cuda, pct).cpu().numpy()
        plain = I.instance_masks_device(inst, h, w, cuda).cpu().numpy()
        for i, (g, a) in enumerate(zip(got, inst)):
            want = cm.ann_to_mask({"segmentation": opb.shrink_instance_masks(a["segmentation"], pct)}, h, w)
            assert np.array_equal(g, want), (pct, i, int((g != want).sum()))
        assert got.sum() < plain.sum()
    calib = I.Calibration(inputs['calib'])
    imgfov = I.map_pointcloud_to_image(inputs['points'], calib, IMG_SHAPE, min_dist=1.0)
    polys = []
    for it in inputs['instances'][:6]:
        x0, y0, bw, bh = it['bbox']
        ang = np.sort(rng.uniform(0, 2 * np.pi, 12))
        pts = np.stack([x0 + bw / 2 + 0.7 * bw * np.cos(ang), y0 + bh / 2 + 0.7 * bh * np.sin(ang)], 1)
        polys.append([float(t) for t in pts.reshape(-1)])
    only_poly = [{"segmentation": [p], "bbox": i["bbox"], "category_id": 1, "box_id": i["box_id"]} for p, i in zip(polys, inputs['instances'])]
    shrunk = I.get_pts_in_mask(None, only_poly, imgfov, shrink_percentage=3, append_mask_info=True)
    with_mask = [dict(a, segmentation=opb.shrink_instance_masks(a["segmentation"], 3)) for a in only_poly]
    with_mask = [dict(a, bin_mask=cm.ann_to_mask(a, *IMG_SHAPE)) for a in with_mask]
    ref = I.get_pts_in_mask(None, [dict(a) for a in with_mask], imgfov, append_mask_info=True)
    full = I.get_pts_in_mask(None, only_poly, imgfov)
    assert len(shrunk["lidar_xyz"]) == len(ref["lidar_xyz"]) > 0
    for la, lb, ia, ib in zip(shrunk["lidar_xyz"], ref["lidar_xyz"], shrunk["img_labels"], ref["img_labels"]):
        assert np.array_equal(la, lb) and np.array_equal(ia["bin_mask"], ib["bin_mask"])
        assert _canonical_rings(ia["segmentation"]) == _canonical_rings(ib["segmentation"])
    assert sum(len(x) for x in shrunk["lidar_xyz"]) < sum(len(x) for x in full["lidar_xyz"])
    assert all(a["segmentation"] == p["segmentation"] for a, p in zip(only_poly, [{"segmentation": [q]} for q in polys]))      # the caller's instances are not modified
