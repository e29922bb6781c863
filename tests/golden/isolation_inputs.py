"""Deterministic inputs for the point-isolation tests: one ray-cast scene with its gt boxes, a KITTI-style
calibration (float32, the dtype Calibration.get_calib_from_file parses to), a 375x1242 image and one binary instance mask per
visible car: an ellipse around the projection of the car's points, which also covers background points behind it -- the
clutter the reference's range-adaptive DBSCAN removes.  Only outputs are stored in isolation.npz; inputs are rebuilt here."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import seevcn_amd.synth as synth  # noqa: E402

IMG_SHAPE = (375, 1242)
PC_ISOLATION = dict(VRES=0.4, EPS_SCALING=4, MAX_EPS=1.0, MIN_EPS=0.0)        # cfgs/KIT-DET_VCN-VC.yaml:14-17
MIN_LIDAR_PTS = 30                                                             # :28
CLASSES = ['Car']


def calibration():
    P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]], np.float32)
    R0 = np.array([[0.9999239, 0.00983776, -0.007445048], [-0.009869795, 0.9999421, -0.004278459],
                   [0.007402527, 0.004351614, 0.9999631]], np.float32)
    V2C = np.array([[0.007533745, -0.9999714, -0.000616602, -0.004069766], [0.01480249, 0.0007280733, -0.9998902, -0.07631618],
                    [0.9998621, 0.00752379, 0.01480755, -0.2717806]], np.float32)
    return {'P2': P2, 'R0': R0, 'Tr_velo2cam': V2C}


def _project(p, calib):
    V2C, R0, P = (calib[k].astype(np.float64) for k in ('Tr_velo2cam', 'R0', 'P2'))
    ref = np.hstack([p[:, :3].astype(np.float64), np.ones((len(p), 1))]) @ V2C.T
    rect = (R0 @ ref.T).T
    img = np.hstack([rect, np.ones((len(p), 1))]) @ P.T
    return img[:, :2] / img[:, 2:3]


def make_inputs(seed=2100, max_instances=10):
    pts, boxes = synth.make_scene(seed, max_boxes=28, box_area=((6.0, 45.0), (-14.0, 14.0)))
    pts = np.ascontiguousarray(pts)
    calib = calibration()
    uv = _project(pts, calib)
    H, W = IMG_SHAPE
    names = np.array(['Car' if int(c) == 1 else ('Pedestrian' if int(c) == 2 else 'Cyclist') for c in boxes[:, 7]])
    yy, xx = np.mgrid[0:H, 0:W]
    instances = []
    for g, b in enumerate(boxes):
        if names[g] != 'Car':
            continue
        c, s = np.cos(b[6]), np.sin(b[6])
        d = pts[:, :3].astype(np.float64) - b[:3]
        lx, ly = d[:, 0] * c + d[:, 1] * s, -d[:, 0] * s + d[:, 1] * c
        inside = (np.abs(lx) <= b[3] / 2) & (np.abs(ly) <= b[4] / 2) & (np.abs(d[:, 2]) <= b[5] / 2)
        inside &= (uv[:, 0] >= 0) & (uv[:, 0] < W) & (uv[:, 1] >= 0) & (uv[:, 1] < H) & (pts[:, 0] > 1.0)
        if inside.sum() < 12:
            continue
        u0, v0 = uv[inside].min(0)
        u1, v1 = uv[inside].max(0)
        cu, cv, ru, rv = (u0 + u1) / 2, (v0 + v1) / 2, (u1 - u0) / 2 * 1.25 + 2, (v1 - v0) / 2 * 1.6 + 2
        mask = (((xx - cu) / ru) ** 2 + ((yy - cv) / rv) ** 2 <= 1.0).astype(np.uint8)
        instances.append({'segmentation': [[float(u0), float(v0), float(u1), float(v0), float(u1), float(v1)]], 'bin_mask': mask,
                          'bbox': [float(u0) - 1.5, float(v0) - 2.5, float(u1 - u0) + 3.7, float(v1 - v0) + 6.2], 'category_id': 1,
                          'box_id': g})
        if len(instances) == max_instances:
            break
    instances.append({'segmentation': [], 'bin_mask': np.zeros(IMG_SHAPE, np.uint8), 'bbox': [0, 0, 1, 1], 'category_id': 1, 'box_id': -1})
    empty = np.zeros(IMG_SHAPE, np.uint8)
    empty[0:3, 0:3] = 1                      # a mask no lidar point falls into: dropped by get_pts_in_mask
    instances.append({'segmentation': [[0, 0, 3, 0, 3, 3]], 'bin_mask': empty, 'bbox': [0, 0, 3, 3], 'category_id': 1, 'box_id': -2})
    sample_infos = {'annos': {'gt_boxes_lidar': boxes[:, :7].copy(), 'name': names, 'num_points_in_gt': np.zeros(len(boxes), np.int64)}}
    return dict(points=pts, boxes=boxes, calib=calib, instances=instances, sample_infos=sample_infos)


def multi_camera_instances(seed=5):
    """Instance clouds as two cameras with overlapping FOVs would produce them: A and A' share points, B is alone, C/C' share
    only two points (below the reference's min_overlap of 3)."""
    rng = np.random.default_rng(seed)
    def blob(centre, n):
        return (rng.normal(0, 0.4, (n, 3)) + np.asarray(centre)).astype(np.float32)
    a = blob([12.0, 3.0, -0.8], 80)
    a2 = np.vstack([a[50:], blob([12.3, 3.2, -0.8], 25)])
    b = blob([30.0, -6.0, -0.7], 60)
    c = blob([20.0, 8.0, -0.9], 50)
    c2 = np.vstack([c[:2], blob([21.5, 8.0, -0.9], 40)])
    return [a, b, c, a2, c2]


def custom_calibration(model):
    """A front camera looking along +x of the lidar (x forward, y left, z up -> camera z forward, x right, y down), lifted 0.3 m, with
    mild distortion; the shapes CustomDatasetObjects.get_calibration returns (intrinsic 3x3, extrinsic 4x4, distcoeff 1x5)."""
    R = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])
    yaw = 0.03
    Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    E = np.eye(4)
    E[:3, :3] = R @ Rz
    E[:3, 3] = [0.05, 0.3, -0.2]
    K = np.array([[960.0, 0.0, 958.3], [0.0, 955.0, 601.7], [0.0, 0.0, 1.0]])
    dist = np.array([[-0.21, 0.07, 0.0012, -0.0007, -0.011]]) if model == "pinhole" else np.array([[0.041, -0.013, 0.0042, -0.0009, 0.0]])
    return {"intrinsic": K, "extrinsic": E, "distcoeff": dist.reshape(-1)}


CUSTOM_IMG_SHAPE = (1208, 1920)
