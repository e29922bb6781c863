"""PCD I/O of the completed clouds -- the wire format between stage A and stage B.

The reference writes them with open3d (`SEE_VCN.save_pcd`, see/surface_completion/SEE_VCN.py:267-280: `o3d.io.write_point_cloud(fname,
pcd, write_ascii=False)`, xyz only) and reads them back in the detector's dataloaders (`get_completed_lidar`,
detector3d/pcdet/datasets/kitti/sc_kitti_dataset.py:20-33: `np.asarray(o3d.io.read_point_cloud(f).points, dtype=np.float32)`).
open3d is an un-vendored dependency; the format itself is PCL's PCD v0.7 and open3d's binary xyz files are

    # .PCD v0.7 - Point Cloud Data file format
    VERSION 0.7 / FIELDS x y z / SIZE 4 4 4 / TYPE F F F / COUNT 1 1 1 / WIDTH n / HEIGHT 1 / VIEWPOINT 0 0 0 1 0 0 0 / POINTS n / DATA binary
    n * 12 bytes of little-endian float32

(the reference's own demo files, demo/demo_data/pcd/*.pcd, pin this: tests/golden/pcd_sample.npz).  write_pcd emits exactly that;
read_pcd accepts any PCD with x, y, z fields of 4- or 8-byte floats, `binary` or `ascii` data (extra fields are skipped) and returns
(N, 3) float32 like the reference's loader.  Host-side I/O: bytes <-> numpy; `to_device=True` hands the points to the GPU in the pcdet
point layout for the next stage."""
import numpy as np

_HEADER = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
           "WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")


def write_pcd(path, points):
    """points (N, >=3) array-like (numpy or torch, any float dtype, host or device) -> binary xyz PCD (float32), open3d's layout."""
    if hasattr(points, "detach"):
        points = points.detach().cpu().numpy()
    xyz = np.ascontiguousarray(np.asarray(points)[:, :3], dtype="<f4")
    with open(path, "wb") as f:
        f.write(_HEADER.format(n=len(xyz)).encode("ascii"))
        f.write(xyz.tobytes())


def read_pcd(path, to_device=None):
    """-> (N, 3) float32 numpy array (or a torch tensor on `to_device`)."""
    with open(path, "rb") as f:
        raw = f.read()
    fields, sizes, types, counts, n_points, data_kind, pos = None, None, None, None, None, None, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", errors="replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, rest = line.partition(" ")
        key = key.upper()
        if key == "FIELDS":
            fields = rest.split()
        elif key == "SIZE":
            sizes = [int(v) for v in rest.split()]
        elif key == "TYPE":
            types = rest.split()
        elif key == "COUNT":
            counts = [int(v) for v in rest.split()]
        elif key == "POINTS":
            n_points = int(rest)
        elif key == "WIDTH" and n_points is None:
            n_points = int(rest)
        elif key == "HEIGHT":
            pass
        elif key == "DATA":
            data_kind = rest.strip().lower()
            break
    if fields is None or sizes is None or types is None or n_points is None:
        raise ValueError(f"{path}: incomplete PCD header")
    counts = counts or [1] * len(fields)
    for ax in "xyz":
        if ax not in fields:
            raise ValueError(f"{path}: PCD without an '{ax}' field")
    if data_kind == "ascii":
        cols, c = {}, 0
        for name, cnt in zip(fields, counts):
            cols[name] = c
            c += cnt
        rows = np.loadtxt(raw[pos:].decode("ascii").splitlines(), dtype=np.float64, ndmin=2) if n_points else np.zeros((0, c))
        xyz = rows[:n_points][:, [cols["x"], cols["y"], cols["z"]]]
    elif data_kind == "binary":
        dt = []
        for name, size, typ, cnt in zip(fields, sizes, types, counts):
            code = {("F", 4): "<f4", ("F", 8): "<f8", ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4", ("I", 8): "<i8",
                    ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4", ("U", 8): "<u8"}.get((typ.upper(), size))
            if code is None:
                raise ValueError(f"{path}: unsupported PCD field type {typ}{size}")
            dt.append((name, code, (cnt,)) if cnt != 1 else (name, code))
        rec = np.frombuffer(raw, dtype=np.dtype(dt), count=n_points, offset=pos)
        xyz = np.stack([rec["x"], rec["y"], rec["z"]], axis=1)
    else:
        raise ValueError(f"{path}: PCD data kind {data_kind!r} is not supported (binary and ascii are)")
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    if to_device is not None:
        import torch
        return torch.from_numpy(xyz).to(to_device)
    return xyz
