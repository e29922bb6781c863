"""Point isolation on the GPU with the reference's names: the reference does this on the CPU with numpy /
open3d right before VCN (see/surface_completion/SEE_VCN.py:61-82,120-209, datasets/shared_utils.py:11-106,201-292,
datasets/kitti/kitti_utils.py:15-114, datasets/kitti/kitti_objects.py:153-176).

  Calibration                     KITTI calibration with the reference's method names, projections on sv_project_lidar_to_image_kitti
  map_pointcloud_to_image         -> imgfov dict (pc_lidar, pc_cam, pts_img, fov_inds, img_shape)
  map_pointcloud_to_image_precomputed / waymo_map_pointcloud_to_image   Waymo: projection made offline, loaded from its .npy pair
  get_pts_in_mask                 -> {"img_uv", "cam_xyz", "lidar_xyz", "img_labels"} lists, binary masks given directly
  isolate_det_pts                 range-adaptive DBSCAN(min_points 3) + largest cluster per instance
  populate_gtboxes / isolate_gt_pts   oriented-box crops of the ground-truth boxes
  merge_multi_camera_detections   stacks instances seen by two cameras

The 2-D segmentation network is out of scope; its output is taken as the reference takes it: instances carry COCO polygons / RLE
dicts under 'segmentation' (rasterised on the device like dataset.annToMask, optionally shrunk first like shrink_instance_masks), a ready binary
mask under 'bin_mask' (what the reference stores after annToMask, shared_utils.py:67-69) or a 'bbox'.
Everything runs through libseevcn_hip.so; there is no CPU fallback.  Results come back as numpy arrays like the reference's,
the *_device variants keep index lists on the GPU.
"""
import numpy as np
import torch

from .. import _lib
from . import polygon_buffer


def _dev_points(points, device):
    if isinstance(points, np.ndarray):
        points = torch.from_numpy(np.ascontiguousarray(points[:, :3], dtype=np.float32))
    points = points.to(device) if points.device.type != 'cuda' else points
    return points.float().contiguous()


def inverse_rigid_trans(Tr):
    inv_Tr = np.zeros_like(Tr)
    inv_Tr[0:3, 0:3] = np.transpose(Tr[0:3, 0:3])
    inv_Tr[0:3, 3] = np.dot(-np.transpose(Tr[0:3, 0:3]), Tr[0:3, 3])
    return inv_Tr


class Calibration(object):
    """kitti_utils.py:15-56: P (3,4), R0 (3,3), V2C (3,4) from a calib dict {'P2','R0','Tr_velo2cam'} or a KITTI calib file."""

    def __init__(self, calib_file):
        calib = calib_file if isinstance(calib_file, dict) else self.get_calib_from_file(calib_file)
        self.P = calib['P2']
        self.R0 = calib['R0']
        self.V2C = calib['Tr_velo2cam']
        self.C2V = inverse_rigid_trans(self.V2C)
        self.c_u, self.c_v = self.P[0, 2], self.P[1, 2]
        self.f_u, self.f_v = self.P[0, 0], self.P[1, 1]
        self.b_x = self.P[0, 3] / (-self.f_u)
        self.b_y = self.P[1, 3] / (-self.f_v)

    def get_calib_from_file(self, calib_file):
        with open(calib_file) as f:
            lines = f.readlines()
        rows = [np.array(lines[i].strip().split(' ')[1:], dtype=np.float32) for i in (2, 3, 4, 5)]
        return {'P2': rows[0].reshape(3, 4), 'P3': rows[1].reshape(3, 4), 'R0': rows[2].reshape(3, 3),
                'Tr_velo2cam': rows[3].reshape(3, 4)}

    def project_device(self, pc_velo, img_w, img_h, min_dist=1.0, device='cuda', want_rect=True):
        """(N,>=3) points -> uv (N,2) int32 (floor; -1 outside), fov (N) bool, rect (N,3) float32 CUDA tensors."""
        lib = _lib.load()
        pts = _dev_points(pc_velo, device)
        n = pts.shape[0]
        uv = torch.empty((n, 2), dtype=torch.int32, device=pts.device)
        fov = torch.empty((n,), dtype=torch.uint8, device=pts.device)
        rect = torch.empty((n, 3), dtype=torch.float32, device=pts.device) if want_rect else None
        v2c = np.ascontiguousarray(self.V2C, dtype=np.float64)
        r0 = np.ascontiguousarray(self.R0, dtype=np.float64)
        p = np.ascontiguousarray(self.P, dtype=np.float64)
        _lib.check(lib.sv_project_lidar_to_image_kitti(_lib.ptr(pts), n, pts.stride(0), v2c.ctypes.data, r0.ctypes.data, p.ctypes.data,
                                                       int(img_w), int(img_h), float(min_dist), _lib.ptr(uv), _lib.ptr(fov),
                                                       _lib.ptr(rect) if rect is not None else None, _lib.stream()),
                   "sv_project_lidar_to_image_kitti")
        return pts, uv, fov.bool(), rect


def map_pointcloud_to_image(pc_velo, calib, img_shape, min_dist=1.0, device='cuda'):
    """KittiObjects.map_pointcloud_to_image (kitti_objects.py:153-176) on arrays: points in the image FOV and farther than
    min_dist along x.  pc_cam is float32 here (the reference's float64 rectified coordinates rounded once)."""
    img_h, img_w = int(img_shape[0]), int(img_shape[1])
    pts, uv, fov, rect = calib.project_device(pc_velo, img_w, img_h, min_dist, device)
    fov_np = fov.cpu().numpy()
    src = pc_velo if isinstance(pc_velo, np.ndarray) else pc_velo.cpu().numpy()
    return {"pc_lidar": src[fov_np, :], "pc_cam": rect[fov].cpu().numpy(), "pts_img": uv[fov].cpu().numpy().astype(int),
            "fov_inds": fov_np, "img_shape": (img_h, img_w), "_device": (pts, uv, fov)}


def map_pointcloud_to_image_custom(points, calib, img_shape, camera_model="pinhole", device='cuda'):
    """CustomDatasetObjects.map_pointcloud_to_image (custom_dataset_objects.py:141-192): calib = {'intrinsic' (3,3), 'extrinsic'
    (>=3,4), 'distcoeff' (5)}, camera_model "pinhole" | "equidistant".  pts_img / pc_cam carry three columns [u, v, depth] like the
    reference's (rounded half-to-even / float64)."""
    if camera_model not in ("pinhole", "equidistant"):
        raise NotImplementedError
    lib = _lib.load()
    img_h, img_w = int(img_shape[0]), int(img_shape[1])
    pts = _dev_points(points, device)
    n = pts.shape[0]
    uvd_int = torch.empty((n, 3), dtype=torch.int32, device=pts.device)
    uvd = torch.empty((n, 3), dtype=torch.float64, device=pts.device)
    fov = torch.empty((n,), dtype=torch.uint8, device=pts.device)
    e = np.ascontiguousarray(np.asarray(calib['extrinsic'], dtype=np.float64)[:3, :])
    k = np.ascontiguousarray(calib['intrinsic'], dtype=np.float64)
    d = np.zeros(5, np.float64)
    dc = np.asarray(calib['distcoeff'], dtype=np.float64).reshape(-1)
    d[:min(5, len(dc))] = dc[:5]
    _lib.check(lib.sv_project_lidar_to_image_camera(_lib.ptr(pts), n, pts.stride(0), e.ctypes.data, k.ctypes.data, d.ctypes.data,
                                                    1 if camera_model == "equidistant" else 0, img_w, img_h, _lib.ptr(uvd_int), _lib.ptr(uvd),
                                                    _lib.ptr(fov), _lib.stream()), "sv_project_lidar_to_image_camera")
    fovb = fov.bool()
    fov_np = fovb.cpu().numpy()
    src = points if isinstance(points, np.ndarray) else points.cpu().numpy()
    return {"pc_lidar": src[fov_np, :], "pc_cam": uvd[fovb].cpu().numpy(), "pts_img": uvd_int[fovb].cpu().numpy().astype(int),
            "fov_inds": fov_np, "img_shape": (img_h, img_w), "_device": (pts, uvd_int[:, :2].contiguous(), fovb)}


def map_pointcloud_to_image_precomputed(points, pts_img, fov_inds, img_shape=None, device='cuda'):
    """WaymoObjects.map_pointcloud_to_image (datasets/waymo/waymo_objects.py:170-186) on arrays: the lidar -> image projection was made offline
    (image_lidar_projections/image_pc and /fov_inds, one .npy pair per frame and camera): pts_img (M, 2+) are the pixel coordinates of the M points
    inside the camera's field of view, fov_inds says which points of the frame's cloud they are -- a boolean mask (N) or an index array (M), both
    index `points` the way the reference's `get_pointcloud(idx)[fov_inds, :]` does.  Returns the reference's imgfov dict (pc_cam is None: Waymo
    keeps no camera-frame coordinates) + the device arrays get_pts_in_mask works on; img_shape (H, W) is only needed for use_bbox lookups."""
    pts = _dev_points(points, device)
    n = pts.shape[0]
    src = points if isinstance(points, np.ndarray) else points.cpu().numpy()
    fi = np.asarray(fov_inds)
    pix = np.asarray(pts_img)
    rows = np.nonzero(fi)[0] if fi.dtype == np.bool_ else fi.astype(np.int64).reshape(-1)
    assert pix.ndim == 2 and pix.shape[0] == rows.shape[0], "pts_img and fov_inds describe different numbers of points"
    assert rows.size == 0 or (rows.min() >= 0 and rows.max() < n), "fov_inds points outside the cloud"
    pos = np.full((n,), -1, dtype=np.int64)                      # point index -> row of the FOV-filtered arrays
    pos[rows] = np.arange(rows.shape[0])
    uv_all = np.full((n, 2), -1, dtype=np.int32)
    uv_all[rows] = pix[:, :2].astype(np.int64).astype(np.int32)   # the reference indexes mask[v, u] with these values as they are
    mask = pos >= 0
    out = {"pc_lidar": src[fi, :], "pts_img": pix, "pc_cam": None, "fov_inds": fov_inds, "_fov_pos": pos,
           "_device": (pts, torch.from_numpy(uv_all).to(pts.device), torch.from_numpy(mask).to(pts.device))}
    if img_shape is not None:
        out["img_shape"] = (int(img_shape[0]), int(img_shape[1]))
    return out


def waymo_map_pointcloud_to_image(root_dir, sequence_name, sample_idx, camera_channel, points, img_shape=None, device='cuda'):
    """The file side of WaymoObjects.map_pointcloud_to_image (waymo_objects.py:170-186): loads
    <root>/image_lidar_projections/image_pc/<camera>/<sequence>_<sample:04>.npy and .../fov_inds/<camera>/... and hands them to
    map_pointcloud_to_image_precomputed together with the frame's cloud (`points` = WaymoObjects.get_pointcloud(idx), waymo_objects.py:140-152)."""
    import os
    name = f'{sequence_name}_{int(sample_idx):04}.npy'
    base = os.path.join(str(root_dir), 'image_lidar_projections')
    pts_img = np.load(os.path.join(base, 'image_pc', camera_channel, name))
    fov_inds = np.load(os.path.join(base, 'fov_inds', camera_channel, name))
    return map_pointcloud_to_image_precomputed(points, pts_img, fov_inds, img_shape=img_shape, device=device)


def _quat_to_matrix(q):
    """Rotation matrix of a (w, x, y, z) quaternion, normalised first (what pyquaternion's Quaternion(q).rotation_matrix returns)."""
    w, x, y, z = (np.asarray(q, np.float64) / np.linalg.norm(np.asarray(q, np.float64)))
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)


def map_pointcloud_to_image_nuscenes(points, cs_lidar, pose_lidar, pose_cam, cs_cam, img_shape, min_dist=1.0, device='cuda'):
    """NuScenesObjects.map_pointcloud_to_image (datasets/nuscenes/nuscenes_objects.py:237-295) on arrays: the four nuScenes records the
    reference fetches through the devkit -- calibrated_sensor and ego_pose of the LIDAR_TOP sample_data, ego_pose and calibrated_sensor
    of the camera sample_data, each a dict with 'rotation' (w,x,y,z) and 'translation' (3,), the last one also 'camera_intrinsic' -- are
    passed in directly (no devkit needed).  Same return dict as the reference: pc_lidar, pc_cam (float32 camera frame), pts_img
    (floor(u,v) int), fov_inds, img_shape."""
    lib = _lib.load()
    img_h, img_w = int(img_shape[0]), int(img_shape[1])
    pts = _dev_points(points, device)
    n = pts.shape[0]
    rot = np.ascontiguousarray(np.stack([_quat_to_matrix(cs_lidar['rotation']), _quat_to_matrix(pose_lidar['rotation']),
                                         _quat_to_matrix(pose_cam['rotation']).T, _quat_to_matrix(cs_cam['rotation']).T]), dtype=np.float64)
    tr = np.ascontiguousarray(np.stack([np.asarray(cs_lidar['translation'], np.float64), np.asarray(pose_lidar['translation'], np.float64),
                                        -np.asarray(pose_cam['translation'], np.float64), -np.asarray(cs_cam['translation'], np.float64)]), dtype=np.float64)
    k = np.ascontiguousarray(cs_cam['camera_intrinsic'], dtype=np.float64)
    pc_cam = torch.empty((n, 3), dtype=torch.float32, device=pts.device)
    uv = torch.empty((n, 2), dtype=torch.int32, device=pts.device)
    fov = torch.empty((n,), dtype=torch.uint8, device=pts.device)
    _lib.check(lib.sv_project_lidar_to_image_nuscenes(_lib.ptr(pts), n, pts.stride(0), rot.ctypes.data, tr.ctypes.data, k.ctypes.data, img_w, img_h,
                                                      float(min_dist), _lib.ptr(pc_cam), _lib.ptr(uv), _lib.ptr(fov), _lib.stream()),
               "sv_project_lidar_to_image_nuscenes")
    fovb = fov.bool()
    fov_np = fovb.cpu().numpy()
    src = points if isinstance(points, np.ndarray) else points.cpu().numpy()
    return {"pc_lidar": src[fov_np, :3], "pc_cam": pc_cam[fovb].cpu().numpy(), "pts_img": uv[fovb].cpu().numpy().astype(int),
            "fov_inds": fov_np, "img_shape": (img_h, img_w), "_device": (pts, uv, fovb)}


def points_in_masks_device(uv, fov, masks=None, rects=None, cap=None):
    """uv (N,2) int32, fov (N) bool/uint8, masks (I,H,W) uint8 or rects (I,4) int32 -> index (I,cap) int32 ascending point
    indices, count (I) int32 (CUDA tensors)."""
    lib = _lib.load()
    _lib.require_cuda(uv, fov)
    n = uv.shape[0]
    inst = masks if masks is not None else rects
    n_inst = inst.shape[0]
    cap = int(n if cap is None else cap)
    index = torch.empty((n_inst, max(cap, 1)), dtype=torch.int32, device=uv.device)
    count = torch.zeros((n_inst,), dtype=torch.int32, device=uv.device)
    if masks is not None:
        img_h, img_w = masks.shape[1], masks.shape[2]
    else:
        img_h = img_w = 1 << 30
    fov8 = fov.to(torch.uint8).contiguous()
    _lib.check(lib.sv_points_in_masks(_lib.ptr(uv.contiguous()), _lib.ptr(fov8), n, _lib.ptr(masks) if masks is not None else None,
                                      _lib.ptr(rects) if rects is not None else None, n_inst, img_w, img_h, cap, _lib.ptr(index),
                                      _lib.ptr(count), _lib.stream()), "sv_points_in_masks")
    return index, count


def _rle_counts(counts):
    """the compressed 'counts' string of a COCO RLE dict -> run lengths (pycocotools rleFrString: 5 payload bits per character, bit 5 = more,
    bit 4 of the last character = sign; from the third run on a value is a difference to the run two before)"""
    if isinstance(counts, (list, tuple)):
        return [int(c) for c in counts]
    if isinstance(counts, bytes):
        counts = counts.decode("ascii")
    out, p = [], 0
    while p < len(counts):
        x, k, more = 0, 0, True
        while more:
            c = ord(counts[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(out) > 2:
            x += out[-2]
        out.append(x)
    return out


def _rle_to_mask(seg):
    h, w = seg["size"]
    flat = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for c in _rle_counts(seg["counts"]):
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return np.ascontiguousarray(flat.reshape(w, h).T)                 # the runs are column-major


def instance_masks_device(instances, img_h, img_w, device='cuda', shrink_percentage=0):
    """The binary masks of COCO annotations, (I, img_h, img_w) uint8 on the device: `dataset.annToMask(instance)` of the reference
    (shared_utils.py:66 -> pycocotools annToRLE / frPyObjects / decode).  Polygon segmentations are rasterised by sv_polygons_to_masks (the
    boundary arithmetic of cocoapi's rleFrPoly, union of an instance's parts); RLE dicts are decoded on the host and copied; an instance that
    already carries 'bin_mask' keeps it.

    shrink_percentage != 0 (shrink_instance_masks, shared_utils.py:310-330; polygon instances only, like the reference): every polygon list goes
    through polygon_buffer.shrink_instance_masks first -- the negative buffer's exterior ring(s) with int()-truncated vertices, the original list
    when a part shrinks to nothing -- and the result is what is rasterised, as the reference rasterises its replaced instance['segmentation']
    (vcn/polygon_buffer.py: a restatement of GEOS's construction, unpinned -- shapely is not available here).  Until round 6 this path kept the pixels
    at least d from the part's edges (sv_polygons_to_masks_shrunk: the same region without the vertex list's int() shift; still exported)."""
    lib = _lib.load()
    dev = torch.device(device)
    n = len(instances)
    masks = torch.empty((n, img_h, img_w), dtype=torch.uint8, device=dev)
    xy, off, inst_of, host = [], [0], [], {}
    for i, inst in enumerate(instances):
        seg = inst.get('bin_mask')
        if seg is not None:
            host[i] = np.asarray(seg, np.uint8)
        elif isinstance(inst['segmentation'], dict):
            host[i] = _rle_to_mask(inst['segmentation'])
        else:
            polys = inst['segmentation']
            if shrink_percentage:
                polys = polygon_buffer.shrink_instance_masks(polys, shrink_percentage)
            for poly in polys:
                if len(poly) < 2:
                    continue
                part = [float(t) for t in poly[:2 * (len(poly) // 2)]]
                xy.extend(part)
                off.append(len(xy) // 2)
                inst_of.append(i)
    n_poly = len(inst_of)
    if n_poly:
        xy_t = torch.tensor(xy, dtype=torch.float64, device=dev)
        off_t = torch.tensor(off, dtype=torch.int32, device=dev)
        inst_t = torch.tensor(inst_of, dtype=torch.int32, device=dev)
        max_v = max(b - a for a, b in zip(off[:-1], off[1:]))
        scratch = torch.empty((lib.sv_polygon_masks_scratch_bytes(n_poly, img_h, img_w),), dtype=torch.uint8, device=dev)
    else:
        xy_t = off_t = inst_t = scratch = None
        max_v = 0
    _lib.check(lib.sv_polygons_to_masks(_lib.ptr(xy_t), _lib.ptr(off_t), _lib.ptr(inst_t), n_poly, max_v, n, img_h, img_w, _lib.ptr(scratch),
                                        _lib.ptr(masks), _lib.stream()), "sv_polygons_to_masks")
    for i, m in host.items():
        assert m.shape == (img_h, img_w), f"instance {i}: mask {m.shape} on a {(img_h, img_w)} image"
        masks[i].copy_(torch.from_numpy(np.ascontiguousarray(m)))
    return masks


def instance_masks_within_distance(instances, img_h, img_w, device='cuda', shrink_percentage=0):
    """Round 5's form of the shrunken masks, kept for sv_polygons_to_masks_shrunk's callers: the pixels of every polygon part whose centre is at least
    d (= half diagonal of the part's bounding box x percentage / 100) from the part's edges -- the REGION of the negative buffer without the vertex
    list's int() truncation (get_pts_in_mask takes instance_masks_device / polygon_buffer instead).  An instance with a part that leaves no pixel
    keeps its unshrunken mask.  Polygon instances only."""
    lib = _lib.load()
    dev = torch.device(device)
    n = len(instances)
    masks = torch.zeros((n, img_h, img_w), dtype=torch.uint8, device=dev)
    xy, off, inst_of, dist = [], [0], [], []
    for i, inst in enumerate(instances):
        for poly in inst['segmentation']:
            if len(poly) < 2:
                continue
            part = [float(t) for t in poly[:2 * (len(poly) // 2)]]
            xy.extend(part)
            off.append(len(xy) // 2)
            inst_of.append(i)
            xs, ys = part[0::2], part[1::2]
            dist.append(0.5 * float(np.hypot(max(xs) - min(xs), max(ys) - min(ys))) * (shrink_percentage / 100.0))
    n_poly = len(inst_of)
    if not n_poly:
        return masks
    xy_t = torch.tensor(xy, dtype=torch.float64, device=dev)
    off_t = torch.tensor(off, dtype=torch.int32, device=dev)
    inst_t = torch.tensor(inst_of, dtype=torch.int32, device=dev)
    max_v = max(b - a for a, b in zip(off[:-1], off[1:]))
    scratch = torch.empty((lib.sv_polygon_masks_scratch_bytes(n_poly, img_h, img_w),), dtype=torch.uint8, device=dev)
    dist_t = torch.tensor(dist, dtype=torch.float64, device=dev)
    kept = torch.empty((n_poly,), dtype=torch.int32, device=dev)
    _lib.check(lib.sv_polygons_to_masks_shrunk(_lib.ptr(xy_t), _lib.ptr(off_t), _lib.ptr(inst_t), _lib.ptr(dist_t), n_poly, max_v, n, img_h, img_w,
                                               _lib.ptr(scratch), _lib.ptr(masks), _lib.ptr(kept), _lib.stream()), "sv_polygons_to_masks_shrunk")
    empty = sorted({inst_of[p] for p in np.nonzero(kept.cpu().numpy() == 0)[0]})
    if empty:
        redo = instance_masks_device([{'segmentation': instances[i]['segmentation']} for i in empty], img_h, img_w, dev, 0)
        masks[torch.tensor(empty, device=dev)] = redo
    return masks


def get_pts_in_mask(dataset, instances, imgfov, shrink_percentage=0, use_bbox=False, append_mask_info=False):
    """shared_utils.py:36-106.  Instances may carry COCO polygons / RLE dicts under 'segmentation' (rasterised on the device: instance_masks_device,
    the role of `dataset.annToMask`; the image size comes from dataset.imgs when the COCO object is given, else from imgfov['img_shape']) or a ready
    'bin_mask'.  shrink_percentage != 0 (SHRINK_MASK_PERCENTAGE of the reference's cfgs; shapely's Polygon.buffer(-d) in shared_utils.py:295-330)
    replaces every polygon list by its shrunken one before the masks are made (vcn/polygon_buffer.py: the negative buffer's exterior rings, vertices
    truncated with int(); unpinned -- shapely is not available here).  Like the reference, shrinking applies to polygon segmentations."""
    pts, uv, fov = imgfov["_device"]
    kept = [dict(inst) for inst in instances if inst['segmentation']]
    out = {"img_uv": [], "cam_xyz": [], "lidar_xyz": [], "img_labels": []}
    if not kept:
        return out
    if use_bbox:
        img_h, img_w = imgfov["img_shape"]                       # the reference builds its box mask with this shape (a Waymo imgfov has none)
        rects = []
        for inst in kept:
            bbox = np.array(inst['bbox'])
            bbox[2:4] = bbox[0:2] + bbox[2:4]
            # numpy slicing boxmask[int(y0):int(y1), int(x0):int(x1)] clips to the image
            rects.append([max(int(bbox[0]), 0), max(int(bbox[1]), 0), min(int(bbox[2]), img_w), min(int(bbox[3]), img_h)])
        index, count = points_in_masks_device(uv, fov, rects=torch.tensor(rects, dtype=torch.int32, device=uv.device))
    else:
        if all('bin_mask' in inst for inst in kept) and not shrink_percentage:
            masks = torch.from_numpy(np.ascontiguousarray(np.stack([inst['bin_mask'] for inst in kept]).astype(np.uint8))).to(uv.device)
        else:
            img = getattr(dataset, 'imgs', {}).get(kept[0].get('image_id')) if dataset is not None else None
            img_h, img_w = (img['height'], img['width']) if img else imgfov["img_shape"]
            if shrink_percentage:
                # shared_utils.py:63-64: the instance's polygon list is REPLACED by the shrunken one (the label that is returned carries it, and
                # append_mask_info counts its parts), then rasterised again
                kept = [{k: v for k, v in inst.items() if k != 'bin_mask' or not isinstance(inst['segmentation'], list)} for inst in kept]
                for inst in kept:
                    if isinstance(inst['segmentation'], list):
                        inst['segmentation'] = polygon_buffer.shrink_instance_masks(inst['segmentation'], shrink_percentage)
            masks = instance_masks_device(kept, int(img_h), int(img_w), uv.device, 0)
            masks_h = masks.cpu().numpy()
            for g, inst in enumerate(kept):                           # the reference keeps the binary mask with the label (shared_utils.py:69)
                inst.setdefault('bin_mask', masks_h[g])
        index, count = points_in_masks_device(uv, fov, masks=masks)
    count_h = count.cpu().numpy()
    index_h = index.cpu().numpy()
    fov_pos = imgfov.get("_fov_pos")
    if fov_pos is None:
        fov_pos = np.cumsum(imgfov["fov_inds"]) - 1           # point index -> row of the FOV-filtered arrays
    for g, inst in enumerate(kept):
        if count_h[g] == 0:
            continue
        rows = np.sort(fov_pos[index_h[g, :count_h[g]]])      # the reference walks the FOV-filtered arrays in THEIR order (an index-array fov_inds need not ascend)
        lidar = imgfov["pc_lidar"][rows, :]
        if imgfov["pc_cam"] is not None:
            out["cam_xyz"].append(imgfov["pc_cam"][rows, :])
        if append_mask_info:
            lidar = np.hstack((lidar, inst['category_id'] * np.ones((lidar.shape[0], 1)),
                               inst['segmentation'].__len__() * np.ones((lidar.shape[0], 1))))
        out["lidar_xyz"].append(lidar)
        out["img_uv"].append(imgfov["pts_img"][rows, :])
        out["img_labels"].append(inst)
    return out


def largest_clusters_device(points, starts, counts, max_points, point_index=None, vres=None, eps_scaling=1.0, min_eps=0.0,
                            max_eps=float('inf'), fixed_eps=None, min_points=3, min_cluster=0):
    """One launch for all instances.  points (ΣN,>=3) float32 CUDA; starts (I) int64, counts (I) int32.  Returns out_local
    (same slots as the input lists) int32, out_count (I) int32, eps (I) float64."""
    lib = _lib.load()
    _lib.require_cuda(points, starts, counts)
    n_inst = counts.shape[0]
    slots = point_index.shape[0] if point_index is not None else points.shape[0]
    out_local = torch.empty((max(slots, 1),), dtype=torch.int32, device=points.device)
    out_count = torch.zeros((n_inst,), dtype=torch.int32, device=points.device)
    out_eps = torch.zeros((n_inst,), dtype=torch.float64, device=points.device)
    nbytes = lib.sv_isolate_cluster_scratch_bytes(n_inst, int(max_points))
    scratch = _lib.workspace.scratch("isolate_cluster", nbytes, points.device) if nbytes else None
    tan_vres = float(np.tan(vres * np.pi / 180)) if vres is not None else 0.0
    _lib.check(lib.sv_isolate_largest_cluster(_lib.ptr(points), points.stride(0), _lib.ptr(point_index) if point_index is not None else None,
                                              _lib.ptr(starts), _lib.ptr(counts), n_inst, int(max_points), tan_vres, float(eps_scaling),
                                              float(min_eps), float(max_eps), -1.0 if fixed_eps is None else float(fixed_eps),
                                              int(min_points), int(min_cluster), _lib.ptr(scratch) if scratch is not None else None,
                                              _lib.ptr(out_local), _lib.ptr(out_count), _lib.ptr(out_eps), _lib.stream()),
               "sv_isolate_largest_cluster")
    return out_local, out_count, out_eps


def isolate_det_pts(in_proj_dict, vres, eps_scaling, min_eps, max_eps, min_cluster=10, device='cuda'):
    """SEE_VCN.isolate_det_pts (SEE_VCN.py:144-181): in_proj_dict = list (one per camera) of get_pts_in_mask dicts; returns the
    list of clustered instance point arrays (rows of the input 'lidar_xyz' arrays, input order)."""
    proj_dict = {}
    for key in in_proj_dict[0].keys():
        for pd in in_proj_dict:
            proj_dict.setdefault(key, [])
            proj_dict[key].extend(pd[key])
    clouds = proj_dict['lidar_xyz']
    if not clouds:
        return []
    counts_h = np.array([c.shape[0] for c in clouds], dtype=np.int32)
    starts_h = np.concatenate([[0], np.cumsum(counts_h[:-1], dtype=np.int64)]).astype(np.int64)
    pts = torch.from_numpy(np.ascontiguousarray(np.vstack([c[:, :3] for c in clouds]), dtype=np.float32)).to(device)
    out_local, out_count, _ = largest_clusters_device(pts, torch.from_numpy(starts_h).to(device), torch.from_numpy(counts_h).to(device),
                                                      int(counts_h.max()), vres=vres, eps_scaling=eps_scaling, min_eps=min_eps,
                                                      max_eps=max_eps, min_points=3, min_cluster=min_cluster)
    out_local, out_count = out_local.cpu().numpy(), out_count.cpu().numpy()
    instances = []
    for g, xyz in enumerate(clouds):
        if out_count[g] > min_cluster:
            instances.append(xyz[out_local[starts_h[g]:starts_h[g] + out_count[g]]])
    return instances


def db_scan(points, eps, min_pts=3, return_largest_cluster=False, device='cuda'):
    """shared_utils.db_scan(..., return_largest_cluster=True) (:395-404) on an (N,3) array: the largest cluster, or all points
    when everything is noise.  The list-of-all-clusters form is not provided."""
    assert return_largest_cluster, "only the largest-cluster form is built"
    pts = _dev_points(points, device)
    n = pts.shape[0]
    out_local, out_count, _ = largest_clusters_device(pts, torch.zeros(1, dtype=torch.int64, device=pts.device),
                                                      torch.tensor([n], dtype=torch.int32, device=pts.device), n, fixed_eps=eps,
                                                      min_points=min_pts, min_cluster=0)
    c = int(out_count.item())
    return points if c == 0 else points[out_local[:c].cpu().numpy()]


def gtbox_to_corners(box):
    """shared_utils.py:201-231."""
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


class OrientedBox:
    """The three fields of open3d's OrientedBoundingBox the reference touches (center, R, extent)."""

    def __init__(self, center, R, extent):
        self.center, self.R, self.extent = np.asarray(center, np.float64), np.asarray(R, np.float64), np.asarray(extent, np.float64)

    def get_center(self):
        return self.center

    def row(self):
        return np.concatenate([self.center, self.R.reshape(-1), self.extent])


def get_o3dbox(anno_info, classes):
    """shared_utils.py:274-292.  open3d's create_from_points orders the extent of the (axis-aligned) corner offsets by
    descending size; the reference keeps that extent and overwrites centre and R."""
    gt_box, class_name, num_lidar_pts = anno_info
    if class_name not in classes:
        return None, None, None
    corners, r_mat = gtbox_to_corners(gt_box)
    dims = corners.max(axis=0) - corners.min(axis=0)
    extent = dims[np.argsort(-dims, kind="stable")]
    return OrientedBox(gt_box[0:3], r_mat, extent), num_lidar_pts, gt_box


def populate_gtboxes(sample_infos, dataset_name, classes, add_ground_lift=False, ground_lift_height=0.1):
    """shared_utils.py:11-34."""
    if dataset_name == 'nuscenes':
        zip_infos = zip(sample_infos['gt_boxes'], sample_infos['gt_names'], sample_infos['num_lidar_pts'])
    elif dataset_name in ['kitti', 'waymo', 'custom']:
        anno = sample_infos['annos']
        zip_infos = zip(anno['gt_boxes_lidar'], [name for name in anno['name']], anno['num_points_in_gt'])
    else:
        print(f"{dataset_name} is an unsupported dataset")
        return None
    pcd_gtboxes = {'gt_boxes': [], 'num_lidar_pts': [], 'xyzlwhry_gt_boxes': []}
    for gt_anno in zip_infos:
        box, num_pts, xyzlwhry = get_o3dbox(gt_anno, classes=classes)
        if box is not None:
            if add_ground_lift:
                box.center = box.center + [0, 0, ground_lift_height / 2]
                box.extent = box.extent + [0, 0, -ground_lift_height]
            pcd_gtboxes['gt_boxes'].append(box)
            pcd_gtboxes['num_lidar_pts'].append(num_pts)
            pcd_gtboxes['xyzlwhry_gt_boxes'].append(xyzlwhry)
    return pcd_gtboxes


def crop_boxes_device(points, boxes, cap=None):
    """points (N,>=3) float32 CUDA, boxes list of OrientedBox -> index (G,cap) int32, count (G) int32."""
    lib = _lib.load()
    _lib.require_cuda(points)
    n, G = points.shape[0], len(boxes)
    cap = int(n if cap is None else cap)
    rows = torch.from_numpy(np.ascontiguousarray(np.stack([b.row() for b in boxes]), dtype=np.float64)).to(points.device)
    index = torch.empty((G, max(cap, 1)), dtype=torch.int32, device=points.device)
    count = torch.zeros((G,), dtype=torch.int32, device=points.device)
    _lib.check(lib.sv_crop_points_in_boxes(_lib.ptr(points), n, points.stride(0), _lib.ptr(rows), G, cap, _lib.ptr(index),
                                           _lib.ptr(count), _lib.stream()), "sv_crop_points_in_boxes")
    return index, count


def isolate_gt_pts(pcd_gtboxes, min_lidar_pts, use_seev1=False, device='cuda'):
    """SEE_VCN.isolate_gt_pts (SEE_VCN.py:61-82): pcd_gtboxes['pcd'] is the (N,3) scene array (the reference holds an open3d
    cloud); returns (list of float64 (Ni,3) crops with >= min_lidar_pts points, their labels)."""
    boxes = pcd_gtboxes['gt_boxes']
    if not boxes:
        return [], []
    scene = pcd_gtboxes['pcd']
    pts = _dev_points(scene, device)
    index, count = crop_boxes_device(pts, boxes)
    index, count = index.cpu().numpy(), count.cpu().numpy()
    scene64 = np.asarray(scene if isinstance(scene, np.ndarray) else scene.cpu().numpy(), dtype=np.float64)[:, :3]
    pcds, gt_labels = [], []
    for g in range(len(boxes)):
        if count[g] >= min_lidar_pts:
            pcds.append(scene64[index[g, :count[g]]])
            gt_labels.append(boxes[g] if use_seev1 else pcd_gtboxes['xyzlwhry_gt_boxes'][g])
    return pcds, gt_labels


def merge_multi_camera_detections(isolated_inst, min_overlap=3, min_dist_to_check=3, device='cuda'):
    """SEE_VCN.py:183-209.  The overlap count (points of j within 0.1 m of some point of i, inclusive) runs on
    sv_points_near_set with the threshold moved to the next float64 (strict < next(0.1) == inclusive <= 0.1 on the distance)."""
    lib = _lib.load()
    isolated_inst = list(isolated_inst)
    joined = []
    inst_d = [np.linalg.norm(inst.mean(axis=0)) for inst in isolated_inst]
    n = len(isolated_inst)
    dev = [None] * n
    thresh = float(np.nextafter(0.1, 1.0))
    for i in range(n):
        for j in range(n):
            if (abs(inst_d[i] - inst_d[j]) < min_dist_to_check) & (i != j) & (j not in joined):
                for t in (i, j):
                    if dev[t] is None:
                        dev[t] = _dev_points(np.asarray(isolated_inst[t]), device)
                near = torch.empty((dev[j].shape[0],), dtype=torch.uint8, device=dev[j].device)
                _lib.check(lib.sv_points_near_set(_lib.ptr(dev[j]), dev[j].shape[0], _lib.ptr(dev[i]), dev[i].shape[0], 3, thresh,
                                                  _lib.ptr(near), _lib.stream()), "sv_points_near_set")
                if int(near.sum().item()) > min_overlap:
                    isolated_inst.append(np.vstack([isolated_inst[i], isolated_inst[j]]))
                    joined.extend([i, j])
    return [isolated_inst[i] for i in range(len(isolated_inst)) if i not in joined]
