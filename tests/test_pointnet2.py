"""FPS / ball query / grouping: oracle self-checks (CPU) and HIP vs oracle, index-exact (GPU)."""
import numpy as np
import pytest
import torch

from oracle import pointnet2 as op2


def _bitrev_rule_winner(d2, t):
    """closed form of the reference's tie rule: argmax value, ties -> min (bitreverse(k mod T), k div T)."""
    log2t = int(np.log2(t))
    k = np.arange(len(d2))
    lo = k & (t - 1)
    rev = np.array([int(format(int(x), f"0{log2t}b")[::-1], 2) if log2t else 0 for x in lo])
    order = np.lexsort((k >> log2t, rev, -d2.astype(np.float64)))
    return int(order[0])


def test_oracle_fps_basic_and_tie_rule():
    rng = np.random.default_rng(0)
    xyz = rng.normal(size=(300, 3)).astype(np.float32)
    idx = op2.farthest_point_sampling(xyz, 64)
    assert idx[0] == 0 and len(set(idx.tolist())) == 64
    # greedy property: every pick maximises the distance to the already picked set
    for j in range(1, 10):
        d = ((xyz[:, None] - xyz[idx[:j]][None]) ** 2).sum(-1).min(1)
        assert np.isclose(d[idx[j]], d.max())
    # duplicates force exact ties: the literal thread/tree emulation must agree with the closed-form rule
    dup = np.tile(rng.normal(size=(37, 3)).astype(np.float32), (9, 1))[:300]
    got = op2.farthest_point_sampling(dup, 50)
    temp = np.full(300, 1e10, np.float32)
    old = 0
    for j in range(1, 50):
        d = dup - dup[old]
        dist = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + (d[:, 2] * d[:, 2]).astype(np.float32)
        temp = np.minimum(dist, temp)
        old = _bitrev_rule_winner(temp, 256)
        assert got[j] == old, j


def test_oracle_ball_query_and_group():
    rng = np.random.default_rng(1)
    xyz = rng.uniform(-2, 2, size=(200, 3)).astype(np.float32)
    new = xyz[:20] + 0.01
    idx = op2.ball_query(0.8, 8, xyz, [120, 80], new, [12, 8])
    assert idx.shape == (20, 8)
    for q in range(20):
        lo, hi = (0, 120) if q < 12 else (120, 200)
        d = ((new[q] - xyz[lo:hi]) ** 2).sum(1)
        hits = np.nonzero(d < 0.64)[0]
        assert idx[q, 0] == (hits[0] if len(hits) else -1)
    idx[idx[:, 0] == -1] = 0
    g = op2.group_points(xyz, [120, 80], idx, [12, 8])
    assert g.shape == (20, 3, 8)
    assert np.array_equal(g[15, :, 2], xyz[120 + idx[15, 2]])


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(300, 64), (1024, 128), (5000, 256), (20000, 2048), (1, 1), (7, 7), (30000, 64)])
def test_hip_fps_index_exact(cuda, hip_lib, n, m):
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as pu
    rng = np.random.default_rng(n)
    xyz = rng.normal(size=(2, n, 3)).astype(np.float32) * 10
    xyz[1, n // 2:] = xyz[1, : n - n // 2]            # exact duplicates -> ties
    out = pu.farthest_point_sample(torch.from_numpy(xyz).to(cuda), m).cpu().numpy()
    for b in range(2):
        ref = op2.farthest_point_sampling(xyz[b], m)
        assert np.array_equal(out[b], ref), (b, np.nonzero(out[b] != ref)[0][:5])


@pytest.mark.gpu
def test_hip_stack_fps_matches_per_scene(cuda, hip_lib):
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as pu
    rng = np.random.default_rng(5)
    counts = [4000, 17000, 900]
    xyz = rng.normal(size=(sum(counts), 3)).astype(np.float32) * 20
    idx = pu.stack_farthest_point_sample(torch.from_numpy(xyz).to(cuda), torch.tensor(counts, dtype=torch.int32, device=cuda), 512).cpu().numpy()
    s = 0
    for b, c in enumerate(counts):
        assert np.array_equal(idx[b], op2.farthest_point_sampling(xyz[s:s + c], 512) + s)
        s += c


@pytest.mark.gpu
@pytest.mark.parametrize("counts,m", [([20000, 4097], 1024), ([65536], 300), ([5000, 5000, 5000, 5000], 2048)])
def test_hip_stack_fps_several_workgroups_per_scene_index_exact(cuda, hip_lib, counts, m):
    """Scenes of >= 4096 points take the 16-workgroups-per-scene kernel (records exchanged every round): same picks as the oracle,
    including exact duplicates that tie on distance (tie rule of sampling_gpu.cu:16-21 lives in the key)."""
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as pu
    rng = np.random.default_rng(len(counts) + m)
    xyz = rng.normal(size=(sum(counts), 3)).astype(np.float32) * 20
    xyz[counts[0] // 2:counts[0]] = xyz[:counts[0] - counts[0] // 2]          # duplicates inside scene 0
    for _ in range(2):                                                         # twice: stale records of the first call must not match
        idx = pu.stack_farthest_point_sample(torch.from_numpy(xyz).to(cuda), torch.tensor(counts, dtype=torch.int32, device=cuda), m, None, False).cpu().numpy()
        s = 0
        for b, c in enumerate(counts):
            ref = op2.farthest_point_sampling(xyz[s:s + c], m) + s
            assert np.array_equal(idx[b], ref), (b, np.nonzero(idx[b] != ref)[0][:5])
            s += c


def _fps_cloud(kind, n, rng):
    if kind == "gauss":
        return (rng.normal(size=(n, 3)) * 20).astype(np.float32)
    if kind == "sweep":                                   # lidar-like: rings on a ground plane + a few boxes, 70 x 80 x 4 m
        r = rng.uniform(2, 50, n) ** 1.0
        a = rng.uniform(-np.pi, np.pi, n)
        z = np.where(rng.random(n) < 0.8, rng.normal(-1.6, 0.03, n), rng.uniform(-1.6, 2.0, n))
        return np.stack([r * np.cos(a), r * np.sin(a), z], 1).astype(np.float32)
    if kind == "lattice":                                 # integer lattice: masses of exactly equal distances
        g = rng.integers(0, 12, size=(n, 3))
        return g.astype(np.float32)
    if kind == "dupes":                                   # 97 distinct points, each many times: m > distinct ends in all-zero distances
        base = (rng.normal(size=(97, 3)) * 5).astype(np.float32)
        return base[rng.integers(0, 97, n)]
    raise ValueError(kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,counts,m", [("gauss", [2048], 256), ("gauss", [17000, 5000], 700), ("sweep", [17000, 17011], 1024), ("sweep", [24576], 300),
                                           ("lattice", [4096, 3000], 600), ("dupes", [2500], 300), ("gauss", [3000, 0, 1, 70, 9000], 128)])
def test_hip_fps_on_buckets_index_exact(cuda, hip_lib, kind, counts, m):
    """csrc/fps_bucket.hip (Morton-cell order, buckets skipped when their box is farther than their largest running distance) against the
    oracle: same picks in the same order -- ties (lattice, duplicates, more picks than distinct points), ragged batches with an empty scene, a
    one-point scene and a scene smaller than a bucket, the largest scene the path takes."""
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as ps
    rng = np.random.default_rng(len(counts) * 1000 + m)
    xyz = np.concatenate([_fps_cloud(kind, c, rng) for c in counts]) if sum(counts) else np.zeros((0, 3), np.float32)
    assert hip_lib.sv_fps_bucket_applies(len(counts), max(counts), m)
    idx = ps.stack_farthest_point_sampling(torch.from_numpy(xyz).to(cuda), torch.tensor(counts, dtype=torch.int32, device=cuda), m, max(counts)).cpu().numpy()
    s = 0
    for b, c in enumerate(counts):
        if c:
            ref = op2.farthest_point_sampling(xyz[s:s + c], m) + s
            assert np.array_equal(idx[b], ref), (b, np.nonzero(idx[b] != ref)[0][:5], idx[b][:8], ref[:8])
        s += c


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sweep", "gauss"])
def test_hip_fps_on_buckets_equals_the_exhaustive_kernels_at_full_size(cuda, hip_lib, kind):
    """4 scenes x ~17k points -> 4096 keypoints (the PV-RCNN sampling, source-nuscenes/pvrcnn.yaml:111) and the fixed-size layout (64 x 16384 -> 1024):
    the bucket path and the exhaustive kernels pick the same rows."""
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as ps
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as PB
    from seevcn_amd import _lib
    rng = np.random.default_rng(17)
    counts = [17000, 16411, 17000, 20480]
    xyz = torch.from_numpy(np.concatenate([_fps_cloud(kind, c, rng) for c in counts])).to(cuda)
    cnt = torch.tensor(counts, dtype=torch.int32, device=cuda)
    a = ps.stack_farthest_point_sampling(xyz, cnt, 4096, max(counts))
    b = ps.stack_farthest_point_sampling(xyz, cnt, 4096, max(counts), bucketed=False)
    assert torch.equal(a, b)
    h = ps.stack_farthest_point_sampling_async(xyz, cnt, 4096, max(counts))
    assert h.err is None and torch.equal(h.result(), b)
    fixed = torch.from_numpy(np.stack([_fps_cloud(kind, 16384, rng) for _ in range(8)])).to(cuda)
    got = PB.farthest_point_sample(fixed, 1024)
    want = torch.empty_like(got)
    _lib.check(hip_lib.sv_farthest_point_sampling(_lib.ptr(fixed), 8, 16384, 1024, None, _lib.ptr(want), _lib.stream()), "sv_farthest_point_sampling")
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("radius,nsample", [(0.3, 16), (1.5, 32), (6.0, 64), (12.0, 16)])
def test_hip_hashed_ball_query_equals_the_scan(cuda, hip_lib, radius, nsample):
    """sv_ball_query_stack_hashed (cell hash, nsample smallest indices of the hits) == sv_ball_query_stack (the reference's scan in index order),
    element for element: two scenes that share their coordinates (bucket collisions across scenes), balls with thousands of hits (the hit list
    overflows and is reduced), empty balls, queries far outside the cloud."""
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as raw
    pts, _ = synth.make_scene_batch(1, seed=2100, n_az=300)
    xyz1 = np.ascontiguousarray(pts[:, 1:4])
    rng = np.random.default_rng(5)
    xyz = np.concatenate([xyz1, xyz1[rng.permutation(len(xyz1))]]).astype(np.float32)      # scene 1 = scene 0 in another order
    counts = [len(xyz1), len(xyz1)]
    qcnt = [1500, 1300]
    new = np.concatenate([xyz1[rng.integers(0, len(xyz1), q)] + rng.normal(0, 0.2, (q, 3)) for q in qcnt]).astype(np.float32)
    new[3] = [900, -900, 50]
    new[1700] = [-1e4, 0, 0]
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(cuda) if dt is None else torch.tensor(a, dtype=dt, device=cuda)
    args = (2, len(new), radius, nsample, t(new), t(qcnt, torch.int32), t(xyz), t(counts, torch.int32))
    hashed = torch.zeros((len(new), nsample), dtype=torch.int32, device=cuda)
    scanned = torch.zeros((len(new), nsample), dtype=torch.int32, device=cuda)
    assert len(xyz) >= raw.BALL_HASH_MIN_POINTS
    raw.ball_query_wrapper(*args, hashed)
    saved, raw.BALL_HASH_MIN_POINTS = raw.BALL_HASH_MIN_POINTS, 1 << 40
    try:
        raw.ball_query_wrapper(*args, scanned)
    finally:
        raw.BALL_HASH_MIN_POINTS = saved
    assert torch.equal(hashed, scanned)
    assert int((scanned[:, 0] == -1).sum()) >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("radius,nsample", [(0.4, 16), (0.8, 16), (2.4, 32), (0.05, 4)])
def test_hip_ball_query_group_vs_oracle(cuda, hip_lib, radius, nsample):
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as pu
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as raw
    pts, _ = synth.make_scene_batch(3, seed=2000, n_az=40)
    counts = np.bincount(pts[:, 0].astype(int), minlength=3)
    xyz = np.ascontiguousarray(pts[:, 1:4])
    rng = np.random.default_rng(2)
    qcnt = [300, 257, 64]
    new = np.concatenate([xyz[np.cumsum(counts)[b] - counts[b]:np.cumsum(counts)[b]][rng.integers(0, counts[b], q)] + rng.normal(0, 0.05, (q, 3))
                          for b, q in enumerate(qcnt)]).astype(np.float32)
    new[5] = [500, 500, 500]                           # an empty ball
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(cuda) if dt is None else torch.tensor(a, dtype=dt, device=cuda)
    idx_raw = torch.zeros((len(new), nsample), dtype=torch.int32, device=cuda)
    raw.ball_query_wrapper(3, len(new), radius, nsample, t(new), t(qcnt, torch.int32), t(xyz), t(counts.tolist(), torch.int32), idx_raw)
    ref = op2.ball_query(radius, nsample, xyz, counts, new, qcnt)
    assert np.array_equal(idx_raw.cpu().numpy(), ref)
    # through the reference-shaped autograd API, with gradients
    feats = rng.normal(size=(len(xyz), 16)).astype(np.float32)
    f = t(feats).requires_grad_(True)
    qg = pu.QueryAndGroup(radius, nsample, use_xyz=True)
    new_feats, idx = qg(t(xyz), t(counts.tolist(), torch.int32), t(new), t(qcnt, torch.int32), f)
    ref0 = ref.copy()
    empty = ref0[:, 0] == -1
    ref0[empty] = 0
    assert np.array_equal(idx.cpu().numpy(), ref0) and empty[5]
    gx = op2.group_points(xyz, counts, ref0, qcnt) - new[:, :, None]
    gf = op2.group_points(feats, counts, ref0, qcnt)
    gx[empty] = 0
    gf[empty] = 0
    np.testing.assert_array_equal(new_feats.detach().cpu().numpy(), np.concatenate([gx, gf], 1))
    go = rng.normal(size=new_feats.shape).astype(np.float32)
    new_feats.backward(t(go))
    go_f = go[:, 3:].copy()
    go_f[empty] = 0
    np.testing.assert_allclose(f.grad.cpu().numpy(), op2.group_points_grad(go_f, ref0, qcnt, counts, len(xyz)), rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------------------------------ batch layout + 3-NN interpolation
@pytest.mark.gpu
def test_hip_pointnet2_batch_ops_vs_oracle(cuda, hip_lib):
    """pointnet2_batch wrappers (ball query, grouping, gather, FPS, three_nn, three_interpolate and their gradients) against the
    numpy restatement of the reference kernels / torch indexing."""
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as PB
    rng = np.random.default_rng(12)
    B, N, M, C = 2, 700, 90, 5
    xyz = rng.uniform(-3, 3, (B, N, 3)).astype(np.float32)
    txyz = torch.from_numpy(xyz).to(cuda)
    fidx = PB.farthest_point_sample(txyz, M)
    assert np.array_equal(fidx.cpu().numpy(), np.stack([op2.farthest_point_sampling(xyz[b], M) for b in range(B)]))
    new_xyz = PB.gather_operation(txyz.transpose(1, 2).contiguous(), fidx).transpose(1, 2).contiguous()
    assert np.array_equal(new_xyz.cpu().numpy(), np.stack([xyz[b][fidx[b].cpu().numpy()] for b in range(B)]))
    far = new_xyz.clone()
    far[0, 0] = 100.0                                                              # a query with no neighbour keeps zeros
    for radius, ns in [(0.6, 16), (1.5, 8), (0.05, 4)]:
        idx = PB.ball_query(radius, ns, txyz, far)
        assert np.array_equal(idx.cpu().numpy(), op2.ball_query_batch(radius, ns, xyz, far.cpu().numpy()))
    feats = torch.from_numpy(rng.normal(size=(B, C, N)).astype(np.float32)).to(cuda).requires_grad_(True)
    idx = PB.ball_query(0.6, 16, txyz, new_xyz)
    grouped = PB.grouping_operation(feats, idx)
    want = torch.gather(feats.unsqueeze(2).expand(B, C, M, N), 3, idx.long().unsqueeze(1).expand(B, C, M, 16))
    assert torch.equal(grouped, want)
    w = torch.from_numpy(rng.normal(size=tuple(grouped.shape)).astype(np.float32)).to(cuda)
    (g1,) = torch.autograd.grad((grouped * w).sum(), feats)
    (g2,) = torch.autograd.grad((want * w).sum(), feats)
    torch.testing.assert_close(g1, g2, rtol=1e-5, atol=1e-5)
    qa = PB.QueryAndGroup(0.6, 16)(txyz, new_xyz, feats.detach())
    assert qa.shape == (B, 3 + C, M, 16)
    # three_nn / three_interpolate (batch layout)
    unknown = torch.from_numpy(rng.uniform(-3, 3, (B, 333, 3)).astype(np.float32)).to(cuda)
    dist, idx3 = PB.three_nn(unknown, new_xyz)
    for b in range(B):
        d2, i3 = op2.three_nn(unknown[b].cpu().numpy(), new_xyz[b].cpu().numpy())
        assert np.array_equal(idx3[b].cpu().numpy(), i3) and np.array_equal(dist[b].cpu().numpy(), np.sqrt(d2))
    weight = torch.softmax(-dist, dim=2).contiguous()
    known_f = torch.from_numpy(rng.normal(size=(B, C, M)).astype(np.float32)).to(cuda).requires_grad_(True)
    out = PB.three_interpolate(known_f, idx3, weight)
    for b in range(B):
        want_b = op2.three_interpolate(known_f[b].detach().cpu().numpy().T.copy(), idx3[b].cpu().numpy(), weight[b].cpu().numpy()).T
        assert np.array_equal(out[b].detach().cpu().numpy(), want_b)
    gw = torch.from_numpy(rng.normal(size=tuple(out.shape)).astype(np.float32)).to(cuda)
    (g1,) = torch.autograd.grad((out * gw).sum(), known_f)
    ref = torch.zeros_like(known_f)
    for j in range(3):
        ref.scatter_add_(2, idx3[:, :, j].long().unsqueeze(1).expand(B, C, -1), gw * weight[:, :, j].unsqueeze(1))
    torch.testing.assert_close(g1, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("c_in,mlps,nsamples", [(0, [[16, 16], [16, 16]], [16, 16]), (16, [[16, 16], [16, 32]], [16, 32]), (64, [[64, 64], [64, 64]], [16, 32]),
                                                 (128, [[64, 64], [64, 64]], [16, 16])])
def test_hip_fused_set_abstraction_matches_the_unfused_path(cuda, hip_lib, c_in, mlps, nsamples):
    """Eval-mode StackSAModuleMSG: the fused kernel (ball query -> gather + 2-layer MLP on the matrix core + max, no (M, C+3, ns) tensor) against
    the reference-shaped path (QueryAndGroup -> Conv2d / BatchNorm2d / ReLU -> max_pool2d), incl. empty balls and balls with fewer than nsample
    neighbours; element-wise per channel.  (The unfused path is pinned to the reference's own modules by tests/golden/pvrcnn_heads.npz.)"""
    import seevcn_amd.synth as synth
    from tolerances import assert_close_per_channel
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as pm
    pts, _ = synth.make_scene_batch(3, seed=2000, n_az=60)
    counts = np.bincount(pts[:, 0].astype(int), minlength=3)
    xyz = np.ascontiguousarray(pts[:, 1:4])
    rng = np.random.default_rng(c_in + 1)
    qcnt = [700, 513, 64]
    starts = np.cumsum(counts) - counts
    new = np.concatenate([xyz[starts[b]:starts[b] + counts[b]][rng.integers(0, counts[b], q)] + rng.normal(0, 0.3, (q, 3)) for b, q in enumerate(qcnt)]).astype(np.float32)
    new[7] = [500, 500, 500]                                                # an empty ball
    feats = rng.normal(size=(len(xyz), c_in)).astype(np.float32) if c_in else None
    torch.manual_seed(c_in)
    m = pm.StackSAModuleMSG(radii=[0.4, 1.2], nsamples=nsamples, mlps=[[c_in] + list(x) for x in mlps], use_xyz=True, pool_method='max_pool').to(cuda)
    for mod in m.modules():                                                # non-trivial eval-mode statistics
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.uniform_(-0.5, 0.5)
            mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.uniform_(-0.3, 0.3)
    m.eval()
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(cuda) if dt is None else torch.tensor(a, dtype=dt, device=cuda)
    args = (t(xyz), t(counts.tolist(), torch.int32), t(new), t(qcnt, torch.int32))
    f = t(feats) if c_in else None
    with torch.no_grad():
        assert m._fused_ok(0, args[0], args[2], f) and m._fused_ok(1, args[0], args[2], f)
        _, fused = m(*args, features=f)
        saved, pm.FUSED_SA_OFF, pm.TRAIN_SA_OFF = (pm.FUSED_SA_OFF, pm.TRAIN_SA_OFF), True, True
        try:
            _, plain = m(*args, features=f)
        finally:
            pm.FUSED_SA_OFF, pm.TRAIN_SA_OFF = saved
    assert fused.shape == plain.shape == (sum(qcnt), mlps[0][-1] + mlps[1][-1])
    assert_close_per_channel(fused.cpu().numpy(), plain.cpu().numpy(), rtol=1e-3, atol_frac=1e-4, name="fused SA output")
    # the empty ball: relu(BN shift) through both layers, identical rows for every empty query
    assert torch.isfinite(fused).all()
    # gradients needed -> the fused (forward-only) kernel steps aside
    m.train()
    assert not m._fused_ok(0, args[0], args[2], f)


@pytest.mark.gpu
@pytest.mark.parametrize("c_in,mlps,nsamples", [(0, [[16, 16], [16, 16]], [16, 16]), (64, [[64, 64], [32, 64]], [16, 32]), (128, [[64, 64], [64, 64]], [16, 16]),
                                                 (32, [[32, 32], [48, 16]], [32, 16]), (16, [[16, 16], [16, 32]], [16, 32])])
def test_hip_set_abstraction_train_kernels_match_the_conv2d_path(cuda, hip_lib, c_in, mlps, nsamples):
    """Train-mode StackSAModuleMSG on the hand-written kernels (sv_sa_train_forward / _backward: gather -> MFMA MLP with batch-statistics BatchNorm in
    the passes' epilogues -> max, and the backward chain) against the reference-shaped path (QueryAndGroup -> Conv2d / BatchNorm2d / ReLU ->
    max_pool2d over (1, C, M, nsample)): outputs, running statistics, num_batches_tracked, and the gradients of the point features and of every
    weight, element-wise per channel.  Includes an empty ball, balls padded with their first neighbour and negative BatchNorm scales (the
    maximum over the neighbours is then made by the SMALLEST pre-activation)."""
    import copy
    import seevcn_amd.synth as synth
    from tolerances import assert_close_per_channel
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as pm
    pts, _ = synth.make_scene_batch(3, seed=2000, n_az=60)
    counts = np.bincount(pts[:, 0].astype(int), minlength=3)
    xyz = np.ascontiguousarray(pts[:, 1:4])
    rng = np.random.default_rng(c_in + 5)
    qcnt = [700, 513, 64]
    starts = np.cumsum(counts) - counts
    new = np.concatenate([xyz[starts[b]:starts[b] + counts[b]][rng.integers(0, counts[b], q)] + rng.normal(0, 0.3, (q, 3)) for b, q in enumerate(qcnt)]).astype(np.float32)
    new[7] = [500, 500, 500]
    feats = rng.normal(size=(len(xyz), c_in)).astype(np.float32) if c_in else None
    torch.manual_seed(c_in)
    m1 = pm.StackSAModuleMSG(radii=[0.4, 1.2], nsamples=nsamples, mlps=[[c_in] + list(x) for x in mlps], use_xyz=True, pool_method='max_pool').to(cuda).train()
    with torch.no_grad():
        for mod in m1.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.weight[::5] *= -1.0                                    # negative scales
                mod.bias.uniform_(-0.3, 0.3)
    m2 = copy.deepcopy(m1)
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(cuda) if dt is None else torch.tensor(a, dtype=dt, device=cuda)
    args = (t(xyz), t(counts.tolist(), torch.int32), t(new), t(qcnt, torch.int32))
    f1 = t(feats).requires_grad_(True) if c_in else None
    f2 = t(feats).requires_grad_(True) if c_in else None
    assert m1._train_ok(0, args[0], args[2], f1) and m1._train_ok(1, args[0], args[2], f1)
    _, rows = m1(*args, features=f1)
    saved, pm.TRAIN_SA_OFF = pm.TRAIN_SA_OFF, True
    try:
        _, conv = m2(*args, features=f2)
    finally:
        pm.TRAIN_SA_OFF = saved
    assert rows.shape == conv.shape == (sum(qcnt), mlps[0][-1] + mlps[1][-1])
    assert_close_per_channel(rows.detach().cpu().numpy(), conv.detach().cpu().numpy(), rtol=1e-3, atol_frac=1e-4, name="train-kernel SA output")
    for (n1, b1), (_, b2) in zip(m1.named_buffers(), m2.named_buffers()):
        np.testing.assert_allclose(b1.cpu().numpy(), b2.cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=n1)
        assert "num_batches_tracked" not in n1 or int(b1) == int(b2) == 1
    # the same scale as plain torch ops over the same neighbour lists (gathered rows -> matmul -> F.batch_norm(training) -> relu, twice -> max):
    # the fp32 torch reference of the op; every gradient element-wise per channel
    import torch.nn.functional as F
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as raw
    f3 = t(feats).requires_grad_(True) if c_in else None
    leaves, outs = [], []
    for k in range(2):
        gq = m2.groupers[k]
        idx = torch.zeros((sum(qcnt), gq.nsample), dtype=torch.int32, device=cuda)
        raw.ball_query_wrapper(3, sum(qcnt), gq.radius, gq.nsample, args[2], args[3], args[0], args[1], idx)
        row_start = raw._row_start(args[3], args[1], sum(qcnt))
        empty = idx[:, 0] < 0
        rows_i = row_start[:, None].long() + torch.where(empty[:, None], torch.zeros_like(idx), idx).long()
        x = args[0][rows_i] - args[2][:, None, :]
        if c_in:
            x = torch.cat([x, f3[rows_i]], dim=2)
        x = (x * (~empty)[:, None, None]).reshape(-1, c_in + 3)
        for i in (0, 3):
            conv_i, bn_i = m2.mlps[k][i], m2.mlps[k][i + 1]
            ps = [p.detach().clone().requires_grad_(True) for p in (conv_i.weight.reshape(conv_i.out_channels, -1), bn_i.weight, bn_i.bias)]
            leaves += ps
            x = torch.relu(F.batch_norm(x @ ps[0].t(), None, None, ps[1], ps[2], True, 0.1, bn_i.eps))
        outs.append(x.view(sum(qcnt), gq.nsample, -1).max(dim=1)[0])
    plain = torch.cat(outs, dim=1)
    assert_close_per_channel(rows.detach().cpu().numpy(), plain.detach().cpu().numpy(), rtol=1e-3, atol_frac=1e-4, name="train-kernel SA output vs plain torch ops")
    w = torch.from_numpy(rng.normal(size=tuple(rows.shape)).astype(np.float32)).to(cuda)
    (rows * w).sum().backward()
    (conv * w).sum().backward()
    (plain * w).sum().backward()
    for (n1, p1), p3 in zip(m1.named_parameters(), leaves):
        assert_close_per_channel(p1.grad.reshape(p1.shape[0], -1).cpu().numpy(), p3.grad.reshape(p1.shape[0], -1).cpu().numpy(), rtol=1e-3, atol_frac=1e-4, name=n1)
    if c_in:
        assert_close_per_channel(f1.grad.cpu().numpy(), f3.grad.cpu().numpy(), rtol=1e-3, atol_frac=1e-4, name="feature gradient")
    # against the Conv2d path (MIOpen BatchNorm / pooling kernels): a last-bit difference in a pre-activation can flip a ReLU mask or an arg-max at a
    # near-tie there, which moves one gradient contribution and, through the BatchNorm backward sums, shifts every element a little -- both are
    # valid gradients, so this comparison is norm-wise (the element-wise one is the plain-torch reference above)
    def norm_close(a, b, name):
        assert np.linalg.norm(a - b) <= 2e-2 * np.linalg.norm(b) + 1e-6, (name, float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)))
    for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        norm_close(p1.grad.reshape(p1.shape[0], -1).cpu().numpy(), p2.grad.reshape(p2.shape[0], -1).cpu().numpy(), n1)
    if c_in:
        norm_close(f1.grad.cpu().numpy(), f2.grad.cpu().numpy(), "feature gradient vs the Conv2d path")


@pytest.mark.gpu
def test_hip_group_points_grad_with_padded_balls_vs_index_add(cuda, hip_lib):
    """GroupingOperation.backward (one thread per query x channel, first-neighbour repeats summed first) == a plain index_add."""
    from seevcn_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as pu
    rng = np.random.default_rng(9)
    n, c, ns = [500, 300], 20, 16
    qc = [64, 40]
    feats = torch.from_numpy(rng.normal(size=(sum(n), c)).astype(np.float32)).to(cuda).requires_grad_(True)
    idx = np.concatenate([rng.integers(0, n[b], size=(qc[b], ns)) for b in range(2)]).astype(np.int32)
    idx[::3, 5:] = idx[::3, :1]                                                # balls padded with their first neighbour
    idx_t = torch.from_numpy(idx).to(cuda)
    out = pu.grouping_operation(feats, torch.tensor(n, dtype=torch.int32, device=cuda), idx_t, torch.tensor(qc, dtype=torch.int32, device=cuda))
    g = torch.from_numpy(rng.normal(size=tuple(out.shape)).astype(np.float32)).to(cuda)
    out.backward(g)
    rows = idx.astype(np.int64) + np.repeat([0, n[0]], qc)[:, None]
    want = torch.zeros_like(feats)
    want.index_add_(0, torch.from_numpy(rows.reshape(-1)).to(cuda), g.permute(0, 2, 1).reshape(-1, c))
    torch.testing.assert_close(feats.grad, want, rtol=1e-4, atol=1e-4)
