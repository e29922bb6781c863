"""Host wrappers over the voxelisation entry points of libseevcn_hip.so."""
import ctypes

import numpy as np
import torch

from ... import _lib


def _geom(pc_range, voxel_size, grid_size):
    r = _lib.host_array(ctypes.c_float, [float(np.float32(v)) for v in pc_range])
    v = _lib.host_array(ctypes.c_float, [float(np.float32(x)) for x in voxel_size])
    g = _lib.host_array(ctypes.c_int32, [int(x) for x in grid_size])
    return r, v, g


def voxelize_dynamic(points, pc_range, voxel_size, grid_size, batch_size, num_features=None,
                     capacity=None, return_point_to_voxel=False, sync=True):
    """Dynamic voxelisation + per-voxel mean (replaces dynamic_mean_vfe.py:38-76).

    points: (P, 1+C) fp32 cuda [b, x, y, z, ...].  Returns (voxel_features (V,C), voxel_coords (V,4) int32
    [b,z,y,x], point_to_voxel or None).  With sync=False the outputs keep `capacity` rows and a device
    int32 count is returned as a 4th value instead of narrowing (no host sync; graph-capturable).
    """
    lib = _lib.load()
    _lib.require_cuda(points)
    assert points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] >= 4
    points = points.contiguous()
    P, stride = points.shape
    C = stride - 1 if num_features is None else int(num_features)
    cap = max(int(P if capacity is None else capacity), 1)
    dev = points.device
    ncells = int(batch_size) * int(grid_size[0]) * int(grid_size[1]) * int(grid_size[2])
    # one persistent index per grid size: the layout (words | chunk counts | chunk bases) depends on the cell count, and only the
    # words and counts are returned to zero by a call -- the scan output of a call on another grid would alias them
    ws = _lib.workspace.persistent(f"vox_index_{ncells}", lib.sv_index_persistent_bytes(ncells), dev)
    scratch = _lib.workspace.scratch("vox_scratch", lib.sv_voxelize_dynamic_scratch_bytes(P, ncells, cap), dev)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    feats = torch.empty((cap, C), dtype=torch.float32, device=dev)
    p2v = torch.empty((P,), dtype=torch.int32, device=dev) if return_point_to_voxel else None
    nvox = torch.empty((1,), dtype=torch.int32, device=dev)
    r, v, g = _geom(pc_range, voxel_size, grid_size)
    rc = lib.sv_voxelize_dynamic(_lib.ptr(points), P, stride, C, r, v, g, int(batch_size), _lib.ptr(ws),
                                 _lib.ptr(scratch), _lib.ptr(coords), _lib.ptr(feats), _lib.ptr(p2v), cap,
                                 _lib.ptr(nvox), _lib.stream())
    _lib.check(rc, "sv_voxelize_dynamic")
    if not sync:
        return feats, coords, p2v, nvox
    n = _lib.host_int(nvox)
    return feats[:n], coords[:n], p2v


def mean_vfe(voxels, voxel_num_points):
    """MeanVFE arithmetic (mean_vfe.py:25-29) on (V, max_points, C) hard voxels."""
    lib = _lib.load()
    _lib.require_cuda(voxels, voxel_num_points)
    voxels = voxels.contiguous().float()
    nump = voxel_num_points.contiguous().to(torch.int32)
    V, mp, C = voxels.shape
    out = torch.empty((V, C), dtype=torch.float32, device=voxels.device)
    rc = lib.sv_mean_vfe(_lib.ptr(voxels), _lib.ptr(nump), V, mp, C, _lib.ptr(out), _lib.stream())
    _lib.check(rc, "sv_mean_vfe")
    return out


def voxelize_hard(points, xyz_offset, num_features, scene_cnt, pc_range, voxel_size, grid_size, max_points, max_voxels):
    """Hard voxelisation of a batch of scenes in one launch (spconv VoxelGenerator semantics, data_processor.py:115-143).

    points (ΣP, stride) fp32 cuda; scene_cnt: list/1-D tensor of per-scene point counts.
    Returns voxels (B,max_voxels,max_points,C), coords (B,max_voxels,3) [z,y,x], num_points (B,max_voxels), num_voxels (B)."""
    lib = _lib.load()
    _lib.require_cuda(points)
    points = points.contiguous().float()
    dev = points.device
    cnt = torch.as_tensor(scene_cnt, dtype=torch.int32, device="cpu")
    B = int(cnt.numel())
    max_scene = int(cnt.max().item()) if B else 0
    cnt_d = cnt.to(dev)
    start_d = (torch.cumsum(cnt_d, 0, dtype=torch.int32) - cnt_d).contiguous()
    total = int(points.shape[0])
    scratch = _lib.workspace.scratch("hardvox", lib.sv_voxelize_hard_scratch_bytes(B, total, max_scene), dev)
    voxels = torch.empty((B, max_voxels, max_points, num_features), dtype=torch.float32, device=dev)
    coords = torch.empty((B, max_voxels, 3), dtype=torch.int32, device=dev)
    nump = torch.empty((B, max_voxels), dtype=torch.int32, device=dev)
    nvox = torch.empty((B,), dtype=torch.int32, device=dev)
    r, v, g = _geom(pc_range, voxel_size, grid_size)
    rc = lib.sv_voxelize_hard(_lib.ptr(points) if total else None, points.shape[1], int(xyz_offset), int(num_features), _lib.ptr(start_d),
                              _lib.ptr(cnt_d), B, total, max_scene, r, v, g, int(max_points), int(max_voxels), _lib.ptr(scratch),
                              _lib.ptr(voxels), _lib.ptr(coords), _lib.ptr(nump), _lib.ptr(nvox), _lib.stream())
    _lib.check(rc, "sv_voxelize_hard")
    return voxels, coords, nump, nvox


def pillar_decorate(voxels, voxel_num_points, coords, voxel_size, pc_range, use_absolute_xyz=True, with_distance=False):
    """(V,mp,C) pillars -> (V,mp,C+6[+1]) decorated point features (pillar_vfe.py:94-118)."""
    lib = _lib.load()
    _lib.require_cuda(voxels, voxel_num_points, coords)
    voxels = voxels.contiguous().float()
    nump = voxel_num_points.contiguous().to(torch.int32)
    coords = coords.contiguous().to(torch.int32)
    V, mp, C = voxels.shape
    co = (C if use_absolute_xyz else C - 3) + 6 + (1 if with_distance else 0)
    out = torch.empty((V, mp, co), dtype=torch.float32, device=voxels.device)
    vs = _lib.host_array(ctypes.c_float, [float(np.float32(x)) for x in voxel_size])
    rg = _lib.host_array(ctypes.c_float, [float(np.float32(x)) for x in pc_range])
    rc = lib.sv_pillar_decorate(_lib.ptr(voxels) if V else None, _lib.ptr(nump) if V else None, _lib.ptr(coords) if V else None, V, mp, C, vs, rg,
                                int(use_absolute_xyz), int(with_distance), _lib.ptr(out) if V else None, _lib.stream())
    _lib.check(rc, "sv_pillar_decorate")
    return out
