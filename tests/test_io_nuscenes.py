"""SURVEY 8(f) remainders: PCD I/O of completed clouds (SEE_VCN.py:267-280, sc_kitti_dataset.py:20-33) and the nuScenes lidar -> image
projection (nuscenes_objects.py:237-295)."""
import os

import numpy as np
import pytest
import torch

from oracle import isolation as oiso


# ------------------------------------------------------------------------------------------ PCD
def test_pcd_writer_emits_open3d_layout_and_reader_round_trips(golden_dir, tmp_path):
    from seevcn_amd.vcn import pcd_io
    g = np.load(os.path.join(golden_dir, "pcd_sample.npz"))
    pts = np.concatenate([g["head"], g["tail"]])
    f = str(tmp_path / "x.pcd")
    pcd_io.write_pcd(f, np.concatenate([pts.astype(np.float64), np.ones((len(pts), 2))], 1))         # float64 + extra columns in, xyz float32 out
    raw = open(f, "rb").read()
    ref_header = bytes(g["header"]).decode().replace(str(int(g["n_points"])), str(len(pts)))
    assert raw.startswith(ref_header.encode())                                                        # open3d's header, line for line
    assert raw[len(ref_header):len(ref_header) + 3072] == bytes(g["bytes_head"])                      # same bytes as the reference's own file
    back = pcd_io.read_pcd(f)
    assert back.dtype == np.float32 and np.array_equal(back, pts)
    pcd_io.write_pcd(f, torch.from_numpy(pts))                                                        # torch in
    assert np.array_equal(pcd_io.read_pcd(f), pts)
    pcd_io.write_pcd(f, np.zeros((0, 3)))
    assert pcd_io.read_pcd(f).shape == (0, 3)


def test_pcd_reader_parses_other_layouts(tmp_path):
    from seevcn_amd.vcn import pcd_io
    pts = np.random.default_rng(0).normal(size=(37, 3)).astype(np.float32)
    f = str(tmp_path / "a.pcd")
    with open(f, "w") as fh:                                                                          # ascii, extra field first
        fh.write("# .PCD v0.7\nVERSION 0.7\nFIELDS intensity x y z\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 37\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
                 "POINTS 37\nDATA ascii\n")
        for p in pts:
            fh.write("0.5 %r %r %r\n" % (float(p[0]), float(p[1]), float(p[2])))
    assert np.array_equal(pcd_io.read_pcd(f), pts)
    rec = np.zeros(37, dtype=[("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("rgb", "<u4")])                # binary float64 + packed colour
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    with open(f, "wb") as fh:
        fh.write(b"VERSION 0.7\nFIELDS x y z rgb\nSIZE 8 8 8 4\nTYPE F F F U\nCOUNT 1 1 1 1\nWIDTH 37\nHEIGHT 1\nPOINTS 37\nDATA binary\n" + rec.tobytes())
    assert np.array_equal(pcd_io.read_pcd(f), pts)
    with open(f, "wb") as fh:
        fh.write(b"VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary_compressed\n")
    with pytest.raises(ValueError):
        pcd_io.read_pcd(f)


def test_reference_demo_cloud_fixture_is_self_consistent(golden_dir):
    g = np.load(os.path.join(golden_dir, "pcd_sample.npz"))
    assert int(g["n_points"]) == 26715 and g["head"].shape == (256, 3) and np.isfinite(g["checksum"]).all()


# ------------------------------------------------------------------------------------------ nuScenes projection
def _records(rng):
    def quat(yaw, pitch=0.0, roll=0.0):
        cy, sy, cp, sp, cr, sr = np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(roll / 2), np.sin(roll / 2)
        return [cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy]
    cs_lidar = {"rotation": quat(-1.5707, 0.003, -0.01), "translation": [0.943713, 0.0, 1.84023]}
    pose_lidar = {"rotation": quat(0.63, 0.002, 0.004), "translation": [411.3039, 1180.8903, 0.0]}
    pose_cam = {"rotation": quat(0.6312, 0.0021, 0.0038), "translation": [411.4199, 1181.1972, 0.0]}
    # camera looking forward: camera z = ego x, camera x = -ego y, camera y = -ego z
    cs_cam = {"rotation": [0.4998, -0.5030, 0.4997, -0.4974], "translation": [1.70079, 0.0159, 1.5110],
              "camera_intrinsic": [[1266.417, 0.0, 816.267], [0.0, 1266.417, 491.507], [0.0, 0.0, 1.0]]}
    return cs_lidar, pose_lidar, pose_cam, cs_cam


def test_oracle_nuscenes_projection_against_a_float64_homogeneous_chain():
    """The step-wise float32 restatement vs an independent 4x4 float64 composition: same FOV set except at borders, pixels within one."""
    rng = np.random.default_rng(4)
    pts = np.concatenate([rng.uniform(-50, 50, (4000, 2)), rng.uniform(-2, 3, (4000, 1))], 1).astype(np.float32)
    cs_lidar, pose_lidar, pose_cam, cs_cam = _records(rng)
    out = oiso.map_pointcloud_to_image_nuscenes(pts, cs_lidar, pose_lidar, pose_cam, cs_cam, (900, 1600), 1.0)

    def T(rec, inverse=False):
        M = np.eye(4)
        M[:3, :3] = oiso.quaternion_rotation_matrix(rec["rotation"])
        M[:3, 3] = rec["translation"]
        return np.linalg.inv(M) if inverse else M
    M = T(cs_cam, True) @ T(pose_cam, True) @ T(pose_lidar) @ T(cs_lidar)
    cam = (M @ np.concatenate([pts.astype(np.float64), np.ones((len(pts), 1))], 1).T)[:3]
    uv = np.asarray(cs_cam["camera_intrinsic"]) @ cam
    uv = uv[:2] / uv[2]
    fov = (cam[2] > 1.0) & (uv[0] > 0) & (uv[0] < 1600) & (uv[1] > 0) & (uv[1] < 900)
    assert 100 < fov.sum() < 2000 and (fov != out["fov_inds"]).sum() <= 3
    both = fov & out["fov_inds"]
    got = np.zeros((len(pts), 2), int)
    got[out["fov_inds"]] = out["pts_img"]
    assert np.abs(got[both] - np.floor(uv[:, both]).T).max() <= 1
    np.testing.assert_allclose(out["pc_cam"], cam.T[out["fov_inds"]], rtol=0, atol=2e-3)        # float32 storage of ~1e3 m global coordinates


@pytest.mark.gpu
def test_hip_nuscenes_projection_bit_exact_vs_oracle(cuda, hip_lib):
    from seevcn_amd.vcn import isolation as iso
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform(-60, 60, (200000, 2)), rng.uniform(-3, 4, (200000, 1)), rng.uniform(size=(200000, 1))], 1).astype(np.float32)
    cs_lidar, pose_lidar, pose_cam, cs_cam = _records(rng)
    want = oiso.map_pointcloud_to_image_nuscenes(pts, cs_lidar, pose_lidar, pose_cam, cs_cam, (900, 1600), 1.0)
    got = iso.map_pointcloud_to_image_nuscenes(pts, cs_lidar, pose_lidar, pose_cam, cs_cam, (900, 1600), 1.0)
    assert np.array_equal(got["fov_inds"], want["fov_inds"]) and want["fov_inds"].sum() > 10000
    assert np.array_equal(got["pts_img"], want["pts_img"])                                       # pixels bit-exact
    assert np.array_equal(got["pc_cam"], want["pc_cam"]) and got["pc_cam"].dtype == np.float32   # every float32 rounding step reproduced
    assert np.array_equal(got["pc_lidar"], want["pc_lidar"]) and got["img_shape"] == (900, 1600)
    # feeds the same mask lookup as the KITTI chain
    pts_d, uv, fov = got["_device"]
    rects = torch.tensor([[200, 300, 900, 700]], dtype=torch.int32, device=cuda)
    idx, cnt = iso.points_in_masks_device(uv, fov, rects=rects)
    sel = want["pts_img"]
    inside = (sel[:, 0] >= 200) & (sel[:, 0] < 900) & (sel[:, 1] >= 300) & (sel[:, 1] < 700)
    assert int(cnt[0]) == int(inside.sum())
