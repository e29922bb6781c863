"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's point isolation, the CPU step in front of VCN:

  gt_box_to_obb / crop_oriented_box  <- datasets/shared_utils.py:11-34,201-231,274-292 (populate_gtboxes, gtbox_to_corners,
                                        get_o3dbox) + open3d PointCloud.crop(OrientedBoundingBox), SEE_VCN.py:61-82
  project_velo_to_image_kitti        <- datasets/kitti/kitti_utils.py:58-114, datasets/kitti/kitti_objects.py:153-176
  project_custom_camera              <- datasets/custom_dataset/custom_dataset_objects.py:141-192 (pinhole / equidistant distortion)
  pts_in_masks                       <- datasets/shared_utils.py:36-106 (get_pts_in_mask)
  isolate_det_pts                    <- SEE_VCN.py:144-181
  merge_multi_camera_detections      <- SEE_VCN.py:183-209

Pinning: project_velo_to_image_kitti, project_custom_camera and pts_in_masks are pinned bit-exactly against tests/golden/isolation.npz, produced by the
reference's own Calibration / get_pts_in_mask (numpy only) in the build container (tests/golden/make_isolation_golden.py).
isolate_det_pts / isolate_gt_pts / merge_multi_camera_detections are pinned at the glue level by the same fixture: the
reference's own SEE_VCN methods were run with the open3d calls they make (get_center, cluster_dbscan, crop) served by this
oracle.  The open3d algorithms themselves (un-vendored `open3d` pip package, no version pin, not installed): PARITY UNPINNED.
They restate open3d's published code: OrientedBoundingBox::GetPointIndicesWithinBoundingBox (|d . R[:,a]| <= extent[a]/2,
float64), OrientedBoundingBox::CreateFromPoints (PCA of the corner points, extents ordered by descending eigenvalue),
PointCloud::GetCenter (sequential accumulate / n), ClusterDBSCAN (see oracle/postprocess.py).
"""
import numpy as np

from .postprocess import dbscan_labels


def gtbox_to_corners(box):
    """shared_utils.py:201-231: unrotated corner offsets (8,3) and the z-rotation matrix."""
    l, w, h = box[3], box[4], box[5]
    rotation = box[6]
    bounding_box = np.array([
        [-l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2],
        [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2],
        [-h / 2, -h / 2, -h / 2, -h / 2, h / 2, h / 2, h / 2, h / 2]])
    rotation_matrix = np.array([
        [np.cos(rotation), -np.sin(rotation), 0.0],
        [np.sin(rotation), np.cos(rotation), 0.0],
        [0.0, 0.0, 1.0]])
    return bounding_box.transpose(), rotation_matrix


def obb_extent_from_points(corners):
    """open3d OrientedBoundingBox::CreateFromPoints on the 8 UNROTATED corner offsets (get_o3dbox, shared_utils.py:283-288):
    PCA axes sorted by descending eigenvalue, extent = max - min along each.  For axis-aligned corner offsets the PCA axes are
    the coordinate axes, so the extent is (l,w,h) reordered by descending size (stable for ties); the reference then overwrites
    centre and R but keeps this extent."""
    dims = corners.max(axis=0) - corners.min(axis=0)
    order = np.argsort(-dims, kind="stable")
    return dims[order]


def gt_box_to_obb(gt_box, add_ground_lift=True, ground_lift_height=0.1):
    """(centre(3), R(3,3), extent(3)) float64 of the open3d box the reference crops with (populate_gtboxes + get_o3dbox)."""
    gt_box = np.asarray(gt_box)
    corners, r_mat = gtbox_to_corners(gt_box)
    extent = obb_extent_from_points(corners).astype(np.float64)
    center = np.asarray(gt_box[0:3], np.float64).copy()
    if add_ground_lift:
        center = center + [0, 0, ground_lift_height / 2]
        extent = extent + [0, 0, -ground_lift_height]
    return center, np.asarray(r_mat, np.float64), extent


def crop_oriented_box(points, center, R, extent):
    """Indices (ascending) of the points inside the box: open3d GetPointIndicesWithinBoundingBox."""
    p = np.asarray(points, np.float64)[:, :3]
    d = p - center
    dx, dy, dz = R[:, 0], R[:, 1], R[:, 2]
    inside = ((np.abs(d[:, 0] * dx[0] + d[:, 1] * dx[1] + d[:, 2] * dx[2]) <= extent[0] / 2)
              & (np.abs(d[:, 0] * dy[0] + d[:, 1] * dy[1] + d[:, 2] * dy[2]) <= extent[1] / 2)
              & (np.abs(d[:, 0] * dz[0] + d[:, 1] * dz[1] + d[:, 2] * dz[2]) <= extent[2] / 2))
    return np.nonzero(inside)[0]


def isolate_gt_pts(points, gt_boxes, min_lidar_pts, add_ground_lift=True, ground_lift_height=0.1):
    """SEE_VCN.py:61-82 on plain arrays: list of float64 (Ni,3) crops with >= min_lidar_pts points, and their gt boxes."""
    pcds, labels = [], []
    for box in gt_boxes:
        c, R, e = gt_box_to_obb(box, add_ground_lift, ground_lift_height)
        idx = crop_oriented_box(points, c, R, e)
        if len(idx) >= min_lidar_pts:
            pcds.append(np.asarray(points, np.float64)[idx, :3])
            labels.append(box)
    return pcds, labels


def project_velo_to_image_kitti(pc_velo, V2C, R0, P, img_h, img_w, min_dist=1.0):
    """kitti_utils.py:69-114 + kitti_objects.py:160-175 written out per element in float64 (row . column, left to right).
    Returns fov_inds (N) bool, pts_img (Nf,2) int, pc_rect (Nf,3) float64."""
    p = np.asarray(pc_velo, np.float64)[:, :3]
    V2C, R0, P = np.asarray(V2C, np.float64), np.asarray(R0, np.float64), np.asarray(P, np.float64)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    ref = [x * V2C[c, 0] + y * V2C[c, 1] + z * V2C[c, 2] + V2C[c, 3] for c in range(3)]
    rect = [R0[c, 0] * ref[0] + R0[c, 1] * ref[1] + R0[c, 2] * ref[2] for c in range(3)]
    img = [rect[0] * P[c, 0] + rect[1] * P[c, 1] + rect[2] * P[c, 2] + P[c, 3] for c in range(3)]
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v = img[0] / img[2], img[1] / img[2]
    fov = (u < img_w) & (u >= 0) & (v < img_h) & (v >= 0) & (np.asarray(pc_velo)[:, 0] > min_dist)
    pts_img = np.floor(np.stack([u[fov], v[fov]], 1)).astype(int)
    return fov, pts_img, np.stack([r[fov] for r in rect], 1)


def project_custom_camera(points, intrinsic, extrinsic, distcoeff, img_h, img_w, camera_model="pinhole"):
    """custom_dataset_objects.py:141-192 per element in float64.  Returns fov (N) bool, pts_img (Nf,3) int [u,v,depth] rounded
    half-to-even, uv (Nf,3) float64."""
    p = np.asarray(points, np.float64)[:, :3]
    E, K, d = np.asarray(extrinsic, np.float64), np.asarray(intrinsic, np.float64), np.asarray(distcoeff, np.float64).reshape(-1)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    cam = [E[c, 0] * x + E[c, 1] * y + E[c, 2] * z + E[c, 3] for c in range(3)]
    with np.errstate(divide="ignore", invalid="ignore"):
        xn, yn = cam[0] / cam[2], cam[1] / cam[2]
        pre = (cam[2] > 0) & (np.abs(xn) < np.arctan(img_w / img_h))
        r2 = xn * xn + yn * yn
        if camera_model == "equidistant":
            r1 = np.sqrt(r2)
            a0 = np.arctan(r1)
            a2 = a0 * a0
            a4 = a2 * a2
            a1 = a0 * (1 + d[0] * a2 + d[1] * a4 + d[2] * (a4 * a2) + d[3] * (a4 * a4))
            u, v = (a1 / r1) * xn, (a1 / r1) * yn
        elif camera_model == "pinhole":
            td = 1 + d[0] * r2 + d[1] * (r2 * r2) + d[4] * (r2 * r2 * r2)
            u = xn * td + 2 * d[2] * xn * yn + d[3] * (r2 + 2 * (xn * xn))
            v = yn * td + d[2] * (r2 + 2 * (yn * yn)) + 2 * d[3] * xn * yn
        else:
            raise NotImplementedError
        u = K[0, 0] * u + K[0, 2]
        v = K[1, 1] * v + K[1, 2]
        fov = pre & (u > 0) & (u < img_w - 1) & (v > 0) & (v < img_h - 1)
    uv = np.stack([u[fov], v[fov], cam[2][fov]], 1)
    return fov, np.round(uv, 0).astype(int), uv


def quaternion_rotation_matrix(q):
    """pyquaternion's Quaternion(q).rotation_matrix for a (w, x, y, z) quaternion (normalised first, as pyquaternion does): the nuScenes
    records store rotations this way (calibrated_sensor / ego_pose 'rotation')."""
    w, x, y, z = (np.asarray(q, np.float64) / np.linalg.norm(np.asarray(q, np.float64)))
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)


def map_pointcloud_to_image_nuscenes(points, cs_lidar, pose_lidar, pose_cam, cs_cam, img_shape, min_dist=1.0):
    """NuScenesObjects.map_pointcloud_to_image (see/surface_completion/datasets/nuscenes/nuscenes_objects.py:237-295) without the devkit:
    the records are dicts with 'rotation' (w,x,y,z) and 'translation' (3,) (+ 'camera_intrinsic' for the camera), points (N, >=3).
    LidarPointCloud keeps float32 points and rotate() / translate() store back into that array (nuscenes-devkit data_classes.py), so every
    step rounds to float32; view_points(normalize=True) runs in float64.  PARITY UNPINNED w.r.t. the devkit (un-vendored pip dependency,
    not installed): restated from the reference's own lines + the devkit's published PointCloud.rotate / translate / view_points."""
    pc = np.ascontiguousarray(np.asarray(points, np.float32)[:, :3].T)          # (3, N) float32
    pc_lidar = pc.copy()

    def rotate(R):
        pc[:3, :] = np.dot(R, pc[:3, :])

    def translate(t):
        for i in range(3):
            pc[i, :] = pc[i, :] + t[i]

    rotate(quaternion_rotation_matrix(cs_lidar['rotation']))
    translate(np.array(cs_lidar['translation'], np.float64))
    rotate(quaternion_rotation_matrix(pose_lidar['rotation']))
    translate(np.array(pose_lidar['translation'], np.float64))
    translate(-np.array(pose_cam['translation'], np.float64))
    rotate(quaternion_rotation_matrix(pose_cam['rotation']).T)
    translate(-np.array(cs_cam['translation'], np.float64))
    rotate(quaternion_rotation_matrix(cs_cam['rotation']).T)
    depths = pc[2, :]
    K = np.asarray(cs_cam['camera_intrinsic'], np.float64)
    viewpad = np.eye(4)
    viewpad[:3, :3] = K
    pts = np.dot(viewpad, np.concatenate((pc, np.ones((1, pc.shape[1]))))).astype(np.float64)[:3, :]
    pts = pts / pts[2:3, :].repeat(3, 0).reshape(3, pc.shape[1])
    fov = np.ones(depths.shape[0], dtype=bool)
    fov = np.logical_and(fov, depths > min_dist)
    fov = np.logical_and(fov, pts[0, :] > 0)
    fov = np.logical_and(fov, pts[0, :] < img_shape[1])
    fov = np.logical_and(fov, pts[1, :] > 0)
    fov = np.logical_and(fov, pts[1, :] < img_shape[0])
    return {"pc_lidar": pc_lidar[:3, fov].T, "pc_cam": pc[:3, fov].T, "pts_img": np.floor(pts[:2, fov]).astype(int).T, "fov_inds": fov,
            "img_shape": tuple(img_shape[:2])}


def pts_in_masks(pts_img, masks=None, rects=None):
    """get_pts_in_mask (shared_utils.py:36-106): per instance, positions (into the FOV-filtered arrays) of the points whose
    pixel is set; instances without points are dropped by the caller."""
    out = []
    n = len(masks) if masks is not None else len(rects)
    for g in range(n):
        if masks is not None:
            sel = masks[g][pts_img[:, 1], pts_img[:, 0]].astype(bool)
        else:
            x0, y0, x1, y1 = [int(t) for t in rects[g]]
            sel = (pts_img[:, 1] >= y0) & (pts_img[:, 1] < y1) & (pts_img[:, 0] >= x0) & (pts_img[:, 0] < x1)
        out.append(np.nonzero(sel)[0])
    return out


def get_center(xyz):
    """open3d PointCloud::GetCenter: std::accumulate in point order (float64), divided by n."""
    s = np.zeros(3, np.float64)
    for row in np.asarray(xyz, np.float64)[:, :3]:
        s = s + row
    return s / len(xyz)


def instance_eps(xyz, vres, eps_scaling, min_eps, max_eps):
    c = get_center(xyz)
    dist = np.sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2])
    ring_height = dist * np.tan(vres * np.pi / 180)
    return float(np.clip(eps_scaling * ring_height, a_max=max_eps, a_min=min_eps))


def largest_cluster_indices(xyz, eps, min_points=3):
    labels = dbscan_labels(np.asarray(xyz)[:, :3], eps, min_points)
    y = np.bincount(labels[labels >= 0])
    if len(y) == 0:
        return None
    return np.nonzero(labels == np.argmax(y))[0]


def isolate_det_pts(lidar_xyz, vres, eps_scaling, min_eps, max_eps, min_cluster=10):
    """SEE_VCN.py:144-181: per instance range-adaptive DBSCAN(min_points 3), largest cluster, size filters."""
    instances = []
    for xyz in lidar_xyz:
        if xyz.shape[0] > min_cluster:
            eps = instance_eps(xyz, vres, eps_scaling, min_eps, max_eps)
            idx = largest_cluster_indices(xyz, eps, 3)
            if idx is not None and len(idx) > min_cluster:
                instances.append(xyz[idx])
    return instances


def merge_multi_camera_detections(isolated_inst, min_overlap=3, min_dist_to_check=3):
    """SEE_VCN.py:183-209: instances whose mean ranges differ by < min_dist_to_check and that share more than min_overlap
    points of j within 0.1 m (inclusive, cKDTree.query_ball_point) of i are stacked; the pair members are dropped."""
    isolated_inst = list(isolated_inst)
    joined = []
    inst_d = [np.linalg.norm(inst.mean(axis=0)) for inst in isolated_inst]
    n = len(isolated_inst)
    for i in range(n):
        for j in range(n):
            if (abs(inst_d[i] - inst_d[j]) < min_dist_to_check) and (i != j) and (j not in joined):
                a, b = np.asarray(isolated_inst[i], np.float64), np.asarray(isolated_inst[j], np.float64)
                num_overlap = 0
                for q in b:
                    d = a - q
                    num_overlap += bool(np.any(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2] <= 0.1 * 0.1))
                if num_overlap > min_overlap:
                    isolated_inst.append(np.vstack([isolated_inst[i], isolated_inst[j]]))
                    joined.extend([i, j])
    return [isolated_inst[i] for i in range(len(isolated_inst)) if i not in joined]
