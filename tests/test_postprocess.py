"""VCN post-processing: surface selection (np.unique -> k-NN -> CPython set order -> tile), largest
DBSCAN cluster, scene merge.  The surface selection is pinned to the reference's own get_partial_mesh_batch
(tests/golden/vcn_post.npz); clustering is parity-unpinned w.r.t. open3d (absent) and cross-checked against sklearn."""
import os

import numpy as np
import pytest
import torch

from oracle import postprocess as opp
from post_inputs import make_pairs


def test_cpython_set_order_emulation_matches_interpreter():
    rng = np.random.default_rng(0)
    for trial in range(300):
        hi = int(rng.choice([16, 100, 600, 1024]))
        vals = rng.integers(0, hi, int(rng.integers(1, 20000)))
        if trial % 3 == 0:
            vals = np.concatenate([rng.permutation(hi)[:int(rng.integers(1, hi + 1))], vals])
        assert [int(v) for v in set(list(vals))] == opp.cpython_set_order(vals)


def test_oracle_surface_select_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "vcn_post.npz"))
    partial, coarse = make_pairs()
    for k in (30, 5):
        got = opp.get_partial_mesh_batch(partial, coarse, k=k)
        assert got.dtype == np.float32 and np.array_equal(got, g[f"surface_k{k}"])


def test_oracle_dbscan_largest_cluster_vs_sklearn(golden_dir):
    from sklearn.cluster import DBSCAN
    g = np.load(os.path.join(golden_dir, "vcn_post.npz"))
    _, coarse = make_pairs()
    for pc, eps in [(g["surface_k30"][2], 0.4), (g["surface_k5"][0], 0.2), (coarse[2], 0.15), (coarse[5], 0.1)]:
        ret, size = opp.largest_cluster(pc, eps=eps, min_points=2, total_pts=1024)
        labels = DBSCAN(eps=eps, min_samples=2).fit(pc.astype(np.float64)).labels_       # d <= eps vs open3d's d < eps: no ties in this data
        sizes = np.bincount(labels[labels >= 0])
        assert size == sizes.max()
        if (sizes == sizes.max()).sum() == 1:
            assert np.array_equal(ret[:size], pc[labels == np.argmax(sizes)].astype(np.float64))
    with pytest.raises(ValueError):
        opp.largest_cluster(np.arange(30, dtype=np.float32).reshape(10, 3) * 10, eps=0.4, min_points=2)


def test_oracle_replace_with_completed_pts_vs_ckdtree():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(1)
    scene = rng.uniform(-5, 5, (4000, 3)).astype(np.float32)
    inst = np.unique(rng.uniform(-2, 2, (900, 3)).astype(np.float32), axis=0)
    merged, near = opp.replace_with_completed_pts(scene, inst, 0.3)
    d = cKDTree(inst.astype(np.float64)).query(scene.astype(np.float64))[0]
    assert np.array_equal(near, d < 0.3) and 0 < near.sum() < len(scene) and len(merged) == len(inst) + (~near).sum()


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_surface_select_bitexact_vs_reference_golden(golden_dir, cuda, hip_lib):
    from seevcn_amd.vcn.utils import sampling as S
    g = np.load(os.path.join(golden_dir, "vcn_post.npz"))
    partial, coarse = make_pairs()
    p, c = torch.from_numpy(partial).to(cuda), torch.from_numpy(coarse).to(cuda)
    for k in (30, 5):
        out, nsel = S.get_partial_mesh_batch_device(p, c, k=k)
        assert np.array_equal(out.cpu().numpy(), g[f"surface_k{k}"])
        want = [opp.partial_with_kdtree(partial[b], coarse[b], k)[1] for b in range(len(partial))]
        assert nsel.cpu().tolist() == want
    assert np.array_equal(S.get_partial_mesh_batch(p, c, k=30), g["surface_k30"])               # reference-named numpy API
    assert np.array_equal(S.partial_with_KDTree(p[1], c[1], k=5), g["surface_k5"][1])
    # other k / ragged sizes against the oracle (k = 1, k = 64, n = 700 partial points, m = 900 coarse points, 333 surface points)
    for k, n, m, sp in [(1, 1024, 1024, 1024), (64, 1024, 1024, 1024), (20, 700, 900, 333), (3, 1, 17, 50)]:
        out, _ = S.get_partial_mesh_batch_device(p[:, :n].contiguous(), c[:, :m].contiguous(), k=k, surface_pts=sp)
        want = opp.get_partial_mesh_batch(partial[:, :n], coarse[:, :m], k=k, surface_pts=sp)
        assert np.array_equal(out.cpu().numpy(), want), (k, n, m, sp)


@pytest.mark.gpu
def test_hip_largest_cluster_bitexact_vs_oracle(golden_dir, cuda, hip_lib):
    from seevcn_amd.vcn.utils import sampling as S
    g = np.load(os.path.join(golden_dir, "vcn_post.npz"))
    _, coarse = make_pairs()
    for pcs, eps, mp in [(g["surface_k30"], 0.4, 2), (g["surface_k5"], 0.2, 2), (coarse, 0.15, 2), (coarse, 0.1, 1), (coarse[:, :333], 0.12, 2)]:
        out, cnt = S.get_largest_cluster_batch_device(torch.from_numpy(np.ascontiguousarray(pcs)).to(cuda), eps=eps, min_points=mp, total_pts=1024)
        for b in range(len(pcs)):
            want, size = opp.largest_cluster(pcs[b], eps=eps, min_points=mp, total_pts=1024)
            assert int(cnt[b]) == size and np.array_equal(out[b].cpu().numpy().astype(np.float64), want), (eps, mp, b)
    res = S.get_largest_cluster_batch(g["surface_k30"], eps=0.4, min_points=2, total_pts=1024)
    assert res.dtype == np.float64 and res.shape == (8, 1024, 3)
    with pytest.raises(ValueError):                                                                # all noise -> reference raises
        S.get_largest_cluster_batch(np.arange(30, dtype=np.float32).reshape(1, 10, 3) * 10, eps=0.4, min_points=2)
    with pytest.raises(Exception, match="min_points"):
        S.get_largest_cluster_batch(coarse, eps=0.4, min_points=5)


@pytest.mark.gpu
def test_hip_largest_cluster_period_hint_changes_nothing(golden_dir, cuda, hip_lib):
    """The periodic fast path (sv_vcn_largest_cluster_periodic: pair tests over the first period[b] rows, copies under their originals) against the
    all-pairs kernel and the oracle: the surface selection's own output with its n_selected as the period (the pipeline's call), synthetic tiled clouds
    whose period leaves a partial last copy (rows without a copy), and WRONG hints (a cloud that does not repeat, a period of 1 / n / beyond n), which
    must fall back to all pairs -- bit-identical clouds and sizes everywhere."""
    from seevcn_amd.vcn.utils import sampling as S
    g = np.load(os.path.join(golden_dir, "vcn_post.npz"))
    partial, coarse = make_pairs()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    surface, nsel = S.get_partial_mesh_batch_device(dev(partial), dev(coarse), k=30)
    assert int(nsel.max()) < surface.shape[1]                                       # the selection really tiles
    cases = [(surface, nsel, 0.4, 2), (surface, nsel, 0.05, 2), (surface, nsel, 0.3, 1)]
    rng = np.random.default_rng(4)
    for U in (1, 2, 333, 600, 1023):                                                # 600, 1023: rows U .. n - U have no copy
        base = rng.normal(size=(5, U, 3)).astype(np.float32) * 0.5
        tiled = np.tile(base, (1, -(-1024 // U), 1))[:, :1024]
        cases.append((dev(tiled), torch.full((5,), U, dtype=torch.int32, device=cuda), 0.25, 2))
    plain = dev(coarse)
    for wrong in (1, 7, 512, plain.shape[1], plain.shape[1] + 5, 0, -3):                  # hints that do not hold (or are no hints): all-pairs path
        cases.append((plain, torch.full((plain.shape[0],), wrong, dtype=torch.int32, device=cuda), 0.15, 2))
    for pcs, period, eps, mp in cases:
        want, wcnt = S.get_largest_cluster_batch_device(pcs, eps=eps, min_points=mp, total_pts=1024)
        got, gcnt = S.get_largest_cluster_batch_device(pcs, eps=eps, min_points=mp, total_pts=1024, period=period)
        assert torch.equal(got, want) and torch.equal(gcnt, wcnt), (eps, mp, period[:3].tolist())
    o, size = opp.largest_cluster(surface[0].cpu().numpy(), eps=0.4, min_points=2, total_pts=1024)
    got, gcnt = S.get_largest_cluster_batch_device(surface[:1], eps=0.4, min_points=2, total_pts=1024, period=nsel[:1])
    assert int(gcnt[0]) == size and np.array_equal(got[0].cpu().numpy().astype(np.float64), o)


@pytest.mark.gpu
def test_hip_scene_merge_vs_oracle(golden_dir, cuda, hip_lib):
    import seevcn_amd.synth as synth
    from seevcn_amd.vcn import scene_merge as M
    g = np.load(os.path.join(golden_dir, "vcn_post.npz"))
    clustered = opp.get_largest_cluster_batch(g["surface_k30"], eps=0.4, min_points=2)
    # place the (object-frame) clusters into a scene at the first boxes' centres
    scene_pts, boxes = synth.make_scene(2000)
    world = [(clustered[b].astype(np.float32) + boxes[b % len(boxes), :3]).astype(np.float64) for b in range(len(clustered))]
    want_inst = opp.merge_instances(world)
    inst = M.merge_instances_device([torch.from_numpy(w.astype(np.float32)).to(cuda) for w in world])
    assert np.array_equal(inst.cpu().numpy(), want_inst.astype(np.float32))
    want, near = opp.replace_with_completed_pts(scene_pts[:, :3], want_inst, 0.1)
    got = M.replace_with_completed_pts(scene_pts[:, :3], want_inst, 0.1)
    assert 0 < near.sum() < len(scene_pts) and got.dtype == np.float64 and np.array_equal(got, want)
    assert np.array_equal(M.points_near_set(torch.from_numpy(scene_pts[:, :3].copy()).to(cuda), inst, 0.1).cpu().numpy(), near)


@pytest.mark.gpu
def test_hip_batched_scene_completion_vs_per_scene_oracle(golden_dir, cuda, hip_lib):
    """complete_scene_batch_device = one launch over rows [b,x,y,z]; must equal the per-scene reference sequence."""
    import seevcn_amd.synth as synth
    from seevcn_amd.vcn import scene_merge as M
    g = np.load(os.path.join(golden_dir, "vcn_post.npz"))
    clustered = opp.get_largest_cluster_batch(g["surface_k30"], eps=0.4, min_points=2).astype(np.float32)
    pts, boxes = synth.make_scene_batch(3, seed=2000)
    obj_scene = np.array([0, 0, 0, 2, 2, 2, 2, 0], np.float32)                       # scene 1 gets no completed object
    world = np.stack([clustered[b] + boxes[int(obj_scene[b])][b, :3] for b in range(8)]).astype(np.float32)
    out = M.complete_scene_batch_device(torch.from_numpy(pts).to(cuda), torch.from_numpy(world).to(cuda), torch.from_numpy(obj_scene).to(cuda), 0.1)
    out = out.cpu().numpy()
    for sc in range(3):
        mine = out[out[:, 0] == sc][:, 1:]
        objs = [world[b] for b in range(8) if obj_scene[b] == sc]
        scene_xyz = pts[pts[:, 0] == sc][:, 1:4]
        want = opp.replace_with_completed_pts(scene_xyz, opp.merge_instances(objs), 0.1)[0] if objs else scene_xyz
        assert np.array_equal(mine.astype(np.float64), np.asarray(want, np.float64)), sc
    # compact=False: same rows, the replaced scene points stay in place with scene id -1; the voxeliser sees the same scene
    from seevcn_amd.pcdet.ops import voxel_ops
    raw = M.complete_scene_batch_device(torch.from_numpy(pts).to(cuda), torch.from_numpy(world).to(cuda), torch.from_numpy(obj_scene).to(cuda), 0.1,
                                        compact=False)
    kept = raw[raw[:, 0] >= 0].cpu().numpy()
    lex = lambda a: a[np.lexsort(a.T[::-1])]
    assert np.array_equal(lex(kept), lex(out)) and raw.shape[0] > out.shape[0]              # the same SET of rows (copies flagged, not sorted away)
    geo = ([0, -40, -3, 70.4, 40, 1], [0.05, 0.05, 0.1], [1408, 1600, 40])
    fa, ca, _ = voxel_ops.voxelize_dynamic(raw.contiguous(), *geo, 3)
    fb, cb, _ = voxel_ops.voxelize_dynamic(torch.from_numpy(out).to(cuda), *geo, 3)
    assert torch.equal(ca, cb)
    np.testing.assert_allclose(fa.cpu().numpy(), fb.cpu().numpy(), rtol=1e-5, atol=1e-5)
