"""Deterministic synthetic LiDAR inputs for tests and bench (numpy only, no reference code).

Shapes follow SURVEY.md §8(d): car-sized partial surfaces for VCN (config 2) and ray-cast
KITTI-/nuScenes-shaped scenes with boxes on a ground plane for the detector path (configs 3-5).
"""
import numpy as np

CAR_DIMS = (3.9, 1.6, 1.56)  # anchor size of the 'Car' class in the reference's second.yaml


def rotz(points, yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    r = np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])
    return points @ r


def make_object(rng, n_min=30, n_max=400, dims=CAR_DIMS, r_min=5.0, r_max=60.0, noise=0.02):
    """One cropped car instance: points on the two sensor-facing faces of a box.

    Returns (points (Ni,3) float32, gt_box (7,) float32 [x,y,z,dx,dy,dz,yaw]).
    """
    n = int(rng.integers(n_min, n_max + 1))
    rad = rng.uniform(r_min, r_max)
    az = rng.uniform(-np.pi, np.pi)
    yaw = rng.uniform(-np.pi, np.pi)
    centre = np.array([rad * np.cos(az), rad * np.sin(az), -1.73 + dims[2] / 2])
    # direction to the sensor in the box frame picks the visible +/-x and +/-y faces
    to_sensor = rotz(-centre[None, :], -yaw)[0]
    sx = 1.0 if to_sensor[0] >= 0 else -1.0
    sy = 1.0 if to_sensor[1] >= 0 else -1.0
    n_long = int(n * dims[0] / (dims[0] + dims[1]))
    u = rng.uniform(-0.5, 0.5, size=(n, 2))
    pts = np.empty((n, 3))
    # long face (normal +/-y)
    pts[:n_long, 0] = u[:n_long, 0] * dims[0]
    pts[:n_long, 1] = sy * dims[1] / 2
    pts[:n_long, 2] = u[:n_long, 1] * dims[2]
    # short face (normal +/-x)
    pts[n_long:, 0] = sx * dims[0] / 2
    pts[n_long:, 1] = u[n_long:, 0] * dims[1]
    pts[n_long:, 2] = u[n_long:, 1] * dims[2]
    pts = rotz(pts, yaw) + centre + rng.normal(0.0, noise, size=(n, 3))
    box = np.array([*centre, *dims, yaw])
    return pts.astype(np.float32), box.astype(np.float32)


def resample(points, n_points, rng):
    """Tile then pick n_points by permutation (same contract as the reference's ResamplePoints,
    see/surface_completion/models/vcn/datasets/data_transforms.py:254-262, with an explicit rng)."""
    reps = int(np.ceil(n_points / len(points)))
    tiled = np.tile(points, (reps, 1))
    choice = rng.permutation(tiled.shape[0])
    return tiled[choice[:n_points]]


def make_object_batch(n_objects, seed=1000, n_points=1024):
    """(B,1024,3) float32 resampled partial clouds + (B,7) boxes; object i uses rng seed+i."""
    clouds, boxes = [], []
    for i in range(n_objects):
        rng = np.random.default_rng(seed + i)
        p, b = make_object(rng)
        clouds.append(resample(p, n_points, rng))
        boxes.append(b)
    return np.stack(clouds).astype(np.float32), np.stack(boxes).astype(np.float32)


def _ray_box(origins_dirs, centre, dims, yaw):
    """Slab test of unit rays from the origin against one yawed box. Returns t (inf = miss)."""
    d = rotz(origins_dirs, -yaw)
    o = rotz(-centre[None, :], -yaw)[0]
    half = np.asarray(dims) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (-half - o) / d
        t2 = (half - o) / d
    tmin = np.nanmax(np.minimum(t1, t2), axis=1)
    tmax = np.nanmin(np.maximum(t1, t2), axis=1)
    hit = (tmax >= tmin) & (tmin > 0)
    return np.where(hit, tmin, np.inf)


def make_scene(seed, n_beams=64, elev=(-24.8, 2.0), az=(-45.0, 45.0), n_az=313, max_boxes=64,
               box_area=((5.0, 65.0), (-35.0, 35.0)), ground_z=-1.73, dropout=0.05, max_range=80.0,
               sizes=((3.9, 1.6, 1.56), (0.8, 0.6, 1.73), (1.76, 0.6, 1.73)), n_sweeps=1, z_shift=0.0):
    """Ray-cast scene. Returns (points (P,3) float32, gt_boxes (G,8) float32 with class in col 7)."""
    rng = np.random.default_rng(seed)
    n_box = int(rng.integers(max_boxes // 4, max_boxes + 1))
    boxes = []
    tries = 0
    while len(boxes) < n_box and tries < 50 * n_box:
        tries += 1
        cls = int(rng.integers(0, len(sizes)))
        dx, dy, dz = sizes[cls]
        x = rng.uniform(*box_area[0])
        y = rng.uniform(*box_area[1])
        yaw = rng.uniform(-np.pi, np.pi)
        r = 0.5 * np.hypot(dx, dy)
        if any(np.hypot(x - b[0], y - b[1]) < r + 0.5 * np.hypot(b[3], b[4]) + 0.2 for b in boxes):
            continue
        boxes.append([x, y, ground_z + dz / 2, dx, dy, dz, yaw, cls + 1])
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 8)
    clouds = []
    for s in range(n_sweeps):
        el = np.deg2rad(np.linspace(elev[0], elev[1], n_beams))
        a = np.deg2rad(np.linspace(az[0], az[1], n_az, endpoint=(az[1] - az[0]) < 359.0))
        a = a + (rng.uniform(-0.002, 0.002) if n_sweeps > 1 else 0.0)
        ee, aa = np.meshgrid(el, a, indexing="ij")
        d = np.stack([np.cos(ee) * np.cos(aa), np.cos(ee) * np.sin(aa), np.sin(ee)], axis=-1).reshape(-1, 3)
        with np.errstate(divide="ignore"):
            t = np.where(d[:, 2] < 0, ground_z / d[:, 2], np.inf)
        for b in boxes:
            t = np.minimum(t, _ray_box(d, b[:3], b[3:6], b[6]))
        keep = np.isfinite(t) & (t < max_range) & (rng.uniform(size=t.shape) >= dropout)
        pts = d[keep] * t[keep, None] + rng.normal(0.0, 0.01, size=(int(keep.sum()), 3))
        if n_sweeps > 1:
            pts[:, :2] += rng.normal(0.0, 0.05, size=2)
        clouds.append(pts)
    pts = np.concatenate(clouds, axis=0)
    pts[:, 2] += z_shift
    boxes[:, 2] += z_shift
    return pts.astype(np.float32), boxes.astype(np.float32)


def make_scene_batch(batch_size, seed=2000, **kw):
    """Stacked points with batch index in column 0 ((ΣP,4) float32, pcdet collate layout,
    detector3d/pcdet/datasets/dataset.py:187-192) and padded gt_boxes (B,Gmax,8)."""
    pts, boxes = [], []
    for i in range(batch_size):
        p, b = make_scene(seed + i, **kw)
        pts.append(np.concatenate([np.full((len(p), 1), i, np.float32), p], axis=1))
        boxes.append(b)
    gmax = max(len(b) for b in boxes)
    gt = np.zeros((batch_size, gmax, 8), np.float32)
    for i, b in enumerate(boxes):
        gt[i, :len(b)] = b
    return np.concatenate(pts, axis=0), gt
