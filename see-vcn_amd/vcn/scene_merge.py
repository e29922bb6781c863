"""Scene-level merge of completed objects, the tail of SEE_VCN.complete_gt_pts / complete_det_pts
(see/surface_completion/SEE_VCN.py:115,244-265) on the GPU."""
import numpy as np
import torch

from .. import _lib


def merge_instances_device(clustered):
    """np.unique(np.vstack(clustered), axis=0): (B,N,3) or list of (N,3) -> row-sorted unique (M,3) tensor."""
    x = torch.cat(list(clustered), dim=0) if isinstance(clustered, (list, tuple)) else clustered.reshape(-1, 3)
    return torch.unique(x, dim=0)


def points_near_set(query, ref, thresh):
    """bool (Nq,) : query point closer than `thresh` to any ref point (float64 distances).  Rows are [x,y,z] or, for a batch of
    scenes in one launch, [b,x,y,z] (only rows of the same scene are compared)."""
    lib = _lib.load()
    _lib.require_cuda(query, ref)
    q, r = query.detach().float().contiguous(), ref.detach().float().contiguous()
    assert q.dim() == 2 and q.shape[1] == r.shape[1] and q.shape[1] in (3, 4)
    near = torch.empty((q.shape[0],), dtype=torch.uint8, device=q.device)
    scratch = _lib.workspace.scratch("near_set_boxes", lib.sv_points_near_set_scratch_bytes(r.shape[0]), q.device)
    _lib.check(lib.sv_points_near_set_boxed(_lib.ptr(q) if q.numel() else None, q.shape[0], _lib.ptr(r) if r.numel() else None, r.shape[0], q.shape[1],
                                            float(thresh), _lib.ptr(scratch), _lib.ptr(near) if q.numel() else None, _lib.stream()), "sv_points_near_set_boxed")
    return near.bool()


def replace_with_completed_pts_device(points, sc_instances, point_dist_thresh=0.1):
    """points (N,3+) scene cloud, sc_instances (M,3): completed points first, then the scene points farther than the threshold
    from every completed point (only xyz is kept, as the reference's open3d cloud does)."""
    if sc_instances is None:
        return points[:, :3]
    xyz = points[:, :3].contiguous()
    near = points_near_set(xyz, sc_instances, point_dist_thresh)
    return torch.cat([sc_instances.to(xyz.dtype), xyz[~near]], dim=0)


def replace_with_completed_pts(points, sc_instances, point_dist_thresh=0.1, device='cuda'):
    """numpy in / numpy float64 out, like the reference."""
    if sc_instances is None:
        return np.asarray(points)
    p = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(device)
    r = torch.from_numpy(np.ascontiguousarray(sc_instances, dtype=np.float32)).to(device)
    return replace_with_completed_pts_device(p, r, point_dist_thresh).cpu().numpy().astype(np.float64)


def complete_scene_batch_device(points, clustered, object_scene, point_dist_thresh=0.1, compact=True):
    """Batched tail of SEE_VCN.complete_*_pts + replace_with_completed_pts for a whole batch of scenes in one pass:
    points (SP,4) [b,x,y,z], clustered (B_o,N,3) completed objects, object_scene (B_o,) scene id of each object ->
    (SP',4) rows [b,x,y,z]: per-scene unique completed points first, then the scene points not within the threshold of them.
    compact=False keeps the replaced scene points in place with scene id -1 (every consumer downstream -- the voxelisers -- drops
    rows whose scene id is out of range): no boolean-mask compaction, hence no host sync, when the cloud only feeds voxelisation."""
    bcol = object_scene.to(clustered.dtype).view(-1, 1, 1).expand(-1, clustered.shape[1], 1)
    rows = torch.cat([bcol, clustered], dim=2).view(-1, 4)
    if compact:
        inst = torch.unique(rows, dim=0)                          # row-sorted: scene id first, like one np.unique per scene
        near = points_near_set(points, inst, point_dist_thresh)
        return torch.cat([inst, points[~near]], dim=0)
    # the SET of completed points is all the voxelisers need: copies get scene id -1 in place (sv_dedup_rows) instead of a sort, a
    # compaction and the host sync that sizes its result; object-contiguous rows also give k_points_near_set tight tile boxes
    lib = _lib.load()
    rows = rows.float().contiguous()
    n = rows.shape[0]
    scratch = _lib.workspace.scratch("dedup_rows", lib.sv_dedup_rows_scratch_bytes(n), rows.device)
    _lib.check(lib.sv_dedup_rows(_lib.ptr(rows) if n else None, n, _lib.ptr(scratch), _lib.stream()), "sv_dedup_rows")
    near = points_near_set(points, rows, point_dist_thresh)
    out = torch.cat([rows, points], dim=0)
    out[n:, 0].masked_fill_(near, -1.0)
    return out
