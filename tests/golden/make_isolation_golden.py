"""Generate tests/golden/isolation.npz with the REFERENCE's own point-isolation code (see/surface_completion):
  KittiObjects.map_pointcloud_to_image + Calibration (datasets/kitti/kitti_objects.py:153-176, kitti_utils.py)   -- numpy only
  get_pts_in_mask (datasets/shared_utils.py:36-106), mask and bbox mode                                           -- numpy only
  SEE_VCN.isolate_det_pts / isolate_gt_pts / merge_multi_camera_detections (SEE_VCN.py:61-82,144-209) and populate_gtboxes,
  with the open3d calls they make (PointCloud.get_center / cluster_dbscan / crop, OrientedBoundingBox.create_from_points)
  served by oracle/isolation.py + oracle/postprocess.py, because open3d is not installed here (PARITY UNPINNED w.r.t. open3d).
Environment shims, not algorithm changes: np.bool (removed in numpy >= 1.24, used at shared_utils.py:74), stub modules for
cv2 / shapely / pycocotools / the nuScenes, Waymo and custom dataset classes (need their devkits).

Run only in the build container (needs /root/reference):  python tests/golden/make_isolation_golden.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refimport as R  # noqa: E402
from isolation_inputs import (CLASSES, CUSTOM_IMG_SHAPE, IMG_SHAPE, MIN_LIDAR_PTS, PC_ISOLATION, custom_calibration, make_inputs,  # noqa: E402
                              multi_camera_instances)
from oracle import isolation as oiso  # noqa: E402
from oracle.postprocess import dbscan_labels  # noqa: E402

R.import_vcn()
np.bool = bool
import open3d as o3d  # noqa: E402  (the stub module registered by _refimport)


class _PointCloud:
    def __init__(self):
        self.points = np.zeros((0, 3))

    def get_center(self):
        return oiso.get_center(np.asarray(self.points))

    def cluster_dbscan(self, eps, min_points):
        return dbscan_labels(np.asarray(self.points), eps, min_points)

    def crop(self, box):
        out = _PointCloud()
        out.points = np.asarray(self.points)[oiso.crop_oriented_box(np.asarray(self.points), np.asarray(box.center, np.float64),
                                                                    np.asarray(box.R, np.float64), np.asarray(box.extent, np.float64))]
        return out


class _OBB:
    def __init__(self):
        self.center, self.R, self.extent, self.color = np.zeros(3), np.eye(3), np.zeros(3), None

    def create_from_points(self, pts):
        self.extent = oiso.obb_extent_from_points(np.asarray(pts))
        return self


o3d.geometry.PointCloud = _PointCloud
o3d.geometry.OrientedBoundingBox = _OBB
o3d.utility.Vector3dVector = lambda a: np.asarray(a, np.float64)
for name in ("shapely", "shapely.geometry", "pycocotools", "pycocotools.coco"):
    R._mod(name)
sys.modules["shapely"].geometry = sys.modules["shapely.geometry"]
sys.modules["pycocotools.coco"].COCO = object
for name, cls in (("datasets.nuscenes.nuscenes_objects", "NuscenesObjects"), ("datasets.waymo.waymo_objects", "WaymoObjects"),
                  ("datasets.custom_dataset.custom_dataset_objects", "CustomDatasetObjects")):
    R._mod(name, **{cls: object})

from datasets import shared_utils as su  # noqa: E402
from datasets.kitti import kitti_objects, kitti_utils  # noqa: E402
import SEE_VCN as see  # noqa: E402

inp = make_inputs()
pts = inp['points']

ko = object.__new__(kitti_objects.KittiObjects)
ko.get_pointcloud = lambda idx: pts
ko.get_calibration = lambda idx: kitti_utils.Calibration(inp['calib'])
ko.get_image = lambda idx, channel=None: np.zeros(IMG_SHAPE + (3,), np.uint8)
imgfov = ko.map_pointcloud_to_image(0, camera_channel='image_2')
out = {"fov_inds": imgfov["fov_inds"], "pts_img": imgfov["pts_img"].astype(np.int32), "pc_cam": imgfov["pc_cam"]}


class _Masks:
    def annToMask(self, inst):
        return inst['bin_mask']


def _pack(prefix, arrays):
    out[prefix + "_counts"] = np.array([len(a) for a in arrays], np.int64)
    out[prefix + "_rows"] = np.vstack(arrays) if arrays else np.zeros((0, 3))


for mode, use_bbox in (("mask", False), ("bbox", True)):
    proj = su.get_pts_in_mask(_Masks(), inp['instances'], imgfov, shrink_percentage=0, use_bbox=use_bbox)
    _pack(f"inmask_{mode}_lidar", proj["lidar_xyz"])
    _pack(f"inmask_{mode}_uv", proj["img_uv"])
    out[f"inmask_{mode}_box_id"] = np.array([l['box_id'] for l in proj["img_labels"]], np.int64)
    if not use_bbox:
        fake = types.SimpleNamespace(vres=PC_ISOLATION['VRES'], eps_scaling=PC_ISOLATION['EPS_SCALING'], max_eps=PC_ISOLATION['MAX_EPS'],
                                     min_eps=PC_ISOLATION['MIN_EPS'])
        inst = see.SEE_VCN.isolate_det_pts(fake, [proj], min_cluster=10)
        _pack("det_instances", inst)

pcd_gtboxes = su.populate_gtboxes(inp['sample_infos'], 'kitti', CLASSES, add_ground_lift=True, ground_lift_height=0.1)
pcd_gtboxes['pcd'] = su.convert_to_o3dpcd(pts)
fake = types.SimpleNamespace(min_lidar_pts=MIN_LIDAR_PTS, use_seev1=False)
crops, labels = see.SEE_VCN.isolate_gt_pts(fake, pcd_gtboxes)
_pack("gt_crops", crops)
out["gt_labels"] = np.stack(labels) if labels else np.zeros((0, 7))

merged = see.SEE_VCN.merge_multi_camera_detections(types.SimpleNamespace(), multi_camera_instances())
_pack("merged", merged)


# CustomDatasetObjects.map_pointcloud_to_image (custom_dataset_objects.py:141-192): pinhole and equidistant camera models
R._mod("datasets.custom_dataset.custom_dataset_objects_real")
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("custom_dataset_objects_real",
                                              os.path.join(R.REF, "see", "surface_completion", "datasets", "custom_dataset", "custom_dataset_objects.py"))
cdo = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cdo)
for model in ("pinhole", "equidistant"):
    co = object.__new__(cdo.CustomDatasetObjects)
    co.camera_model = model
    co.get_pointcloud = lambda idx: pts
    co.get_image = lambda idx, channel=None: np.zeros(CUSTOM_IMG_SHAPE + (3,), np.uint8)
    co.get_calibration = lambda idx, model=model: custom_calibration(model)
    fovd = co.map_pointcloud_to_image(0, camera_channel="front")
    out[f"custom_{model}_fov_inds"] = fovd["fov_inds"]
    out[f"custom_{model}_pts_img"] = fovd["pts_img"].astype(np.int32)
    out[f"custom_{model}_pc_cam"] = fovd["pc_cam"]
    assert np.array_equal(fovd["pc_lidar"], pts[fovd["fov_inds"]])
np.savez_compressed(os.path.join(HERE, "isolation.npz"), **out)
print({k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "isolation.npz")))
