"""Sparse conv: oracle cross-checked against torch's dense conv3d (CPU); HIP rulebooks bit-exact vs the oracle,
HIP conv forward/backward/dense within 1e-3 rel (GPU).  spconv itself is unpinned third-party (oracle/spconv.py header)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import spconv as osp

RTOL = 1e-3


def _rand_coords(rng, n, batch, shape, clustered=True):
    """Unique voxel coordinates, clustered so that neighbours exist."""
    z = rng.integers(0, shape[0], size=4 * n)
    y = rng.integers(0, shape[1], size=4 * n)
    x = rng.integers(0, shape[2], size=4 * n)
    if clustered:
        y = (y // 3) % max(shape[1] // 2, 1) + shape[1] // 4
        x = (x // 3) % max(shape[2] // 2, 1) + shape[2] // 4
    b = rng.integers(0, batch, size=4 * n)
    c = np.unique(np.stack([b, z, y, x], 1), axis=0)
    c = c[rng.permutation(len(c))[:n]]
    return c.astype(np.int32)


def _ok(a, b, rtol=1e-3, atol_frac=1e-4, name=""):
    """element-wise |a-b| <= rtol*|b| + atol_frac * max|b[:, c]| per channel (tests/tolerances.py)"""
    from tolerances import assert_close_per_channel
    assert_close_per_channel(a, b, rtol=rtol, atol_frac=atol_frac, name=name)
    return True


GEOMS = [  # (ksize, stride, padding)
    (3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0), (3, 1, 1), (2, 2, 0),
]


# ---------------------------------------------------------------------------------- CPU: oracle vs dense conv3d
@pytest.mark.parametrize("ksize,stride,padding", GEOMS)
def test_oracle_sparse_conv_equals_dense_conv3d(ksize, stride, padding):
    rng = np.random.default_rng(0)
    batch, shape, cin, cout = 2, (9, 14, 12), 5, 7
    coords = _rand_coords(rng, 150, batch, shape)
    feats = rng.normal(size=(len(coords), cin)).astype(np.float32)
    k3 = osp._triple(ksize)
    w2x = rng.normal(size=(cout, *k3, cin)).astype(np.float32)
    oc, nbr_out, nbr_in, oshape = osp.rulebook_sparse(coords, shape, ksize, stride, padding)
    out = osp.conv_forward(feats, nbr_out, osp.weight_to_kio(w2x))
    dense_in = torch.from_numpy(osp.dense(feats, coords, batch, shape)).double()
    wt = torch.from_numpy(w2x).permute(0, 4, 1, 2, 3).double()          # (Cout, Cin, kz, ky, kx)
    dense_out = F.conv3d(dense_in, wt, stride=osp._triple(stride), padding=osp._triple(padding)).numpy()
    assert dense_out.shape[2:] == tuple(oshape)
    # values at the active output sites
    got = dense_out[oc[:, 0], :, oc[:, 1], oc[:, 2], oc[:, 3]]
    np.testing.assert_allclose(out, got, rtol=1e-9, atol=1e-9)
    # the output set is exactly the set of sites that see at least one active input
    occ = torch.from_numpy(osp.dense(np.ones((len(coords), 1), np.float32), coords, batch, shape)).double()
    reach = F.conv3d(occ, torch.ones(1, 1, *k3).double(), stride=osp._triple(stride), padding=osp._triple(padding)).numpy()[:, 0] > 0
    mask = np.zeros_like(reach)
    mask[oc[:, 0], oc[:, 1], oc[:, 2], oc[:, 3]] = True
    assert np.array_equal(mask, reach)
    # canonical order = ascending linear key; tables are mutually consistent
    key = ((oc[:, 0].astype(np.int64) * oshape[0] + oc[:, 1]) * oshape[1] + oc[:, 2]) * oshape[2] + oc[:, 3]
    assert (np.diff(key) > 0).all()
    for k in range(nbr_out.shape[0]):
        v = nbr_in[k] >= 0
        assert np.array_equal(nbr_out[k, nbr_in[k, v]], np.nonzero(v)[0])


def test_oracle_subm_equals_dense_conv3d_at_input_sites():
    rng = np.random.default_rng(1)
    batch, shape, cin, cout = 2, (7, 10, 11), 4, 6
    coords = _rand_coords(rng, 120, batch, shape)
    feats = rng.normal(size=(len(coords), cin)).astype(np.float32)
    w2x = rng.normal(size=(cout, 3, 3, 3, cin)).astype(np.float32)
    nbr = osp.rulebook_subm(coords, shape, 3)
    out = osp.conv_forward(feats, nbr, osp.weight_to_kio(w2x))
    dense_in = torch.from_numpy(osp.dense(feats, coords, batch, shape)).double()
    dense_out = F.conv3d(dense_in, torch.from_numpy(w2x).permute(0, 4, 1, 2, 3).double(), padding=1).numpy()
    np.testing.assert_allclose(out, dense_out[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]], rtol=1e-9, atol=1e-9)
    assert np.array_equal(nbr[13], np.arange(len(coords)))  # centre offset maps every row to itself


def test_oracle_backward_matches_autograd_of_dense_conv():
    rng = np.random.default_rng(2)
    batch, shape, cin, cout = 1, (6, 8, 8), 3, 4
    coords = _rand_coords(rng, 60, batch, shape)
    feats = rng.normal(size=(len(coords), cin))
    w2x = rng.normal(size=(cout, 3, 3, 3, cin))
    oc, nbr_out, _, oshape = osp.rulebook_sparse(coords, shape, 3, 2, 1)
    go = rng.normal(size=(len(oc), cout))
    gf, gw = osp.conv_backward(feats, nbr_out, osp.weight_to_kio(w2x), go)
    f = torch.tensor(feats, requires_grad=True)
    w = torch.tensor(w2x, requires_grad=True)
    ci = [torch.tensor(coords[:, i]).long() for i in range(4)]
    dense_in = torch.zeros(batch, *shape, cin, dtype=torch.float64)
    dense_in[ci[0], ci[1], ci[2], ci[3]] = f
    dense_in = dense_in.permute(0, 4, 1, 2, 3)
    o = F.conv3d(dense_in, w.permute(0, 4, 1, 2, 3), stride=2, padding=1)
    sel = o[torch.tensor(oc[:, 0]).long(), :, torch.tensor(oc[:, 1]).long(), torch.tensor(oc[:, 2]).long(), torch.tensor(oc[:, 3]).long()]
    (sel * torch.tensor(go)).sum().backward()
    np.testing.assert_allclose(gf, f.grad.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(gw, osp.weight_to_kio(w.grad.numpy()), rtol=1e-9, atol=1e-9)


def test_shim_api_surface():
    import seevcn_amd.spconv as spconv
    from seevcn_amd.pcdet.models import backbones_3d
    from seevcn_amd.pcdet.utils.spconv_utils import find_all_spconv_keys
    m = backbones_3d.__all__["VoxelBackBone8x"]({}, 3, [1408, 1600, 40])
    assert m.sparse_shape == [41, 1600, 1408] and m.num_point_features == 128
    assert sum(p.numel() for p in m.parameters()) == 711440
    keys = find_all_spconv_keys(m)
    assert "conv_input.0.weight" in keys and "conv_out.0.weight" in keys and len(keys) == 12
    assert m.conv4[0][0].weight.shape == (64, 3, 3, 3, 64) and m.conv_out[0].weight.shape == (128, 3, 1, 1, 64)
    t = spconv.SparseConvTensor(torch.zeros(2, 3), torch.zeros(2, 4, dtype=torch.int32), [4, 4, 4], 1)
    t2 = t.replace_feature(torch.ones(2, 3))
    assert t2.indices is t.indices and t2.indice_dict is t.indice_dict and float(t2.features.sum()) == 6
    r = backbones_3d.__all__["VoxelResBackBone8x"]({}, 5, [1440, 1440, 40])
    assert r.backbone_channels["x_conv4"] == 128


# ---------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("ksize,stride,padding", GEOMS)
def test_hip_sparse_rulebook_bit_exact(cuda, hip_lib, ksize, stride, padding):
    from seevcn_amd.spconv import functional as Fsp
    rng = np.random.default_rng(3)
    batch, shape = 3, (11, 40, 36)
    coords = _rand_coords(rng, 3000, batch, shape)
    oc, nbr_out, nbr_in, oshape = osp.rulebook_sparse(coords, shape, ksize, stride, padding)
    rb = Fsp.build_sparse_rulebook(torch.from_numpy(coords).to(cuda), batch, shape, osp._triple(ksize), osp._triple(stride),
                                   osp._triple(padding))
    torch.cuda.synchronize()
    assert list(rb.out_shape) == list(oshape)
    assert np.array_equal(rb.out_indices.cpu().numpy(), oc)
    assert np.array_equal(rb.nbr_in.cpu().numpy(), nbr_in)
    assert np.array_equal(rb.nbr_out.cpu().numpy(), nbr_out)
    assert np.array_equal(rb.pair_counts().cpu().numpy(), osp.pair_counts(nbr_out))
    # rebuilding on the cleaned persistent index gives the same tables
    rb2 = Fsp.build_sparse_rulebook(torch.from_numpy(coords).to(cuda), batch, shape, osp._triple(ksize), osp._triple(stride),
                                    osp._triple(padding))
    assert np.array_equal(rb2.nbr_out.cpu().numpy(), nbr_out)


@pytest.mark.gpu
def test_hip_subm_rulebook_bit_exact(cuda, hip_lib):
    from seevcn_amd.spconv import functional as Fsp
    rng = np.random.default_rng(4)
    for batch, shape, n, ks in [(2, (41, 160, 140), 6000, (3, 3, 3)), (1, (5, 30, 30), 900, (3, 3, 3)), (2, (9, 20, 20), 500, (1, 3, 3))]:
        coords = _rand_coords(rng, n, batch, shape)          # random row order (like the voxeliser's (b,x,y,z) order)
        nbr = osp.rulebook_subm(coords, shape, ks)
        for cap in (Fsp.CELLMAP_MAX_BYTES, 0):               # dense cell map, then the rank dictionary (grids too large for a map)
            saved, Fsp.CELLMAP_MAX_BYTES = Fsp.CELLMAP_MAX_BYTES, cap
            try:
                for _ in range(2):                            # twice: the persistent workspace must come back clean
                    rb = Fsp.build_subm_rulebook(torch.from_numpy(coords).to(cuda), batch, shape, list(ks))
                    torch.cuda.synchronize()
                    assert np.array_equal(rb.nbr_out.cpu().numpy(), nbr)
            finally:
                Fsp.CELLMAP_MAX_BYTES = saved
        assert np.array_equal(rb.table_for_backward_data().cpu().numpy(), nbr[::-1])
    # empty input
    rb = Fsp.build_subm_rulebook(torch.zeros((0, 4), dtype=torch.int32, device=cuda), 1, (4, 4, 4), [3, 3, 3])
    assert rb.nbr_out.shape == (27, 0)
    rb = Fsp.build_sparse_rulebook(torch.zeros((0, 4), dtype=torch.int32, device=cuda), 1, (4, 4, 4), [3, 3, 3], [2, 2, 2], [1, 1, 1])
    assert rb.n_out == 0


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (64, 64), (64, 128), (128, 128), (128, 64)])
def test_hip_conv_on_a_table_plan_is_bit_identical_to_the_plain_kernels(cuda, hip_lib, cin, cout):
    """Rulebook.plan: regions per XCD, 16-row tiles of equal neighbour-mask class, regrouped row-major table, cost-sorted tile deal, column
    blocks for C_out = 128, submanifold data gradient on the reversed table -- same summation order per output element, so outputs and
    gradients must equal the plain (ungrouped, k-major table) kernels bit for bit."""
    import seevcn_amd.spconv as spconv
    from seevcn_amd.spconv import functional as Fsp
    rng = np.random.default_rng(9)
    batch, shape = 2, (9, 48, 40)
    coords = _rand_coords(rng, 2500, batch, shape)
    feats = rng.normal(size=(len(coords), cin)).astype(np.float32)
    for subm in (True, False):
        res = []
        for planned in (False, True):
            saved, Fsp.USE_PLAN = Fsp.USE_PLAN, planned
            try:
                torch.manual_seed(0)
                conv = (spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="k") if subm
                        else spconv.SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=False)).to(cuda)
                x = spconv.SparseConvTensor(torch.from_numpy(feats).to(cuda).requires_grad_(True), torch.from_numpy(coords).to(cuda), list(shape), batch)
                out = conv(x).features
                if planned:
                    rb = x.indice_dict["k"] if subm else None
                    assert rb is None or ("fwd" in rb._plans and rb.plan("bwd", cout, cin)[3] is True)     # the plan kernel really took the layer
                out.square().sum().backward()
                res.append((out.detach().cpu().numpy(), x.features.grad.cpu().numpy(), conv.weight.grad.cpu().numpy()))
            finally:
                Fsp.USE_PLAN = saved
        for a, b in zip(*res):
            assert np.array_equal(a, b)


@pytest.mark.gpu
def test_hip_weight_fragments_follow_the_weights_even_under_a_fused_optimizer(cuda, hip_lib):
    """The fragment buffers are re-laid on every forward: torch's fused optimisers update parameters WITHOUT bumping their version counter,
    so no version-keyed cache can be trusted (round-2 bug: the bench ran on the fragments of step 1).  Also: buffers are reused per weight
    tensor, never handed to another tensor that happens to get the same device address, and the layout is the documented one."""
    import seevcn_amd.spconv as spconv
    from seevcn_amd.spconv import functional as Fsp
    rng = np.random.default_rng(3)
    coords = _rand_coords(rng, 1500, 1, (9, 32, 32))
    feats = torch.from_numpy(rng.normal(size=(len(coords), 16)).astype(np.float32)).to(cuda)
    conv = spconv.SubMConv3d(16, 32, 3, padding=1, bias=False, indice_key="k").to(cuda)
    opt = torch.optim.SGD(conv.parameters(), lr=0.5, fused=True)

    def run():
        x = spconv.SparseConvTensor(feats, torch.from_numpy(coords).to(cuda), [9, 32, 32], 1)
        return conv(x).features

    y0 = run()
    y0.square().mean().backward()
    v0 = conv.weight._version
    opt.step()
    assert conv.weight._version == v0                              # the trap: the update is invisible to the version counter
    w = osp.weight_to_kio(conv.weight.detach().cpu().numpy())
    want = osp.conv_forward(feats.cpu().numpy(), osp.rulebook_subm(coords, (9, 32, 32), 3), w)
    y1 = run().detach().cpu().numpy()
    assert _ok(y1, want, name="forward after a fused optimiser step") and not np.allclose(y1, y0.detach().cpu().numpy())
    f0 = Fsp.fragment_cache.get(conv.weight_kio())
    f1 = Fsp.fragment_cache.get(conv.weight_kio())
    assert f1[0] is f0[0] and f1[1] is f0[1]                       # same buffers for the same weight tensor
    wt = torch.randn(27, 16, 16, device=cuda)
    ptr = wt.data_ptr()
    Fsp.fragment_cache.get(wt)
    del wt
    w2 = torch.randn(27, 16, 16, device=cuda)                       # usually re-uses the freed block
    b = Fsp.fragment_cache.get(w2)[0]
    # fragment order of the forward view Wt[k][n = c_out][c = c_in]: float4 unit ((k*KQ + q)*NT + t)*64 + lane holds 4 consecutive c_in
    t6 = w2.permute(0, 2, 1).contiguous().view(27, 1, 16, 1, 4, 4)   # (k, t, li, q, kk, 4) with NT = KQ = 1
    assert torch.equal(b, t6.permute(0, 3, 1, 4, 2, 5).reshape(-1)), "stale fragments" if w2.data_ptr() == ptr else "fragment layout"


@pytest.mark.gpu
@pytest.mark.parametrize("fused,channels", [(True, 64), (False, 64), (True, 32), (False, 32)])
def test_hip_table_plan_is_a_permutation_into_regions_with_balanced_waves(cuda, hip_lib, fused, channels, monkeypatch):
    """(fused: the one-launch builder k_plan_region; otherwise the four separate kernels.)  Structure of a plan at bench size: every row appears exactly once, inside its own region; tile_of covers every tile exactly once per
    region; rows of a tile share their mask class in >= 85 % of the tiles; the busiest wave has <= 1.15x the mean work.  channels 32: four tiles per wave on
    a submanifold table -- the tiles a wave works on at a time are neighbours in the cost-sorted list (round 6: they share most offsets)."""
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet.ops import voxel_ops
    from seevcn_amd.spconv import functional as Fsp
    bs = 8 if channels == 64 else 16                          # four tiles per wave are dealt as neighbours from four rounds of units on (the bench's size)
    pts, _ = synth.make_scene_batch(bs, seed=2000, **({} if channels == 64 else {"n_az": 384}))
    feats, coords, _ = voxel_ops.voxelize_dynamic(torch.from_numpy(pts).to(cuda), [0, -40, -3, 70.4, 40, 1], [0.05, 0.05, 0.1], [1408, 1600, 40], bs)
    monkeypatch.setattr(Fsp, "FUSED_PLAN", fused)
    rb = Fsp.build_subm_rulebook(coords, bs, [41, 1600, 1408], [3, 3, 3])                                # ~120 k rows: 7-8 quads per CU bin
    n = rb.n_out
    tp, tile_of, g, rev = rb.plan("fwd", channels, channels)
    assert rev is False and g == (4 if channels == 32 else 2)
    n_pad = (n + 15) // 16 * 16
    rows = tp.perm.cpu().numpy()
    assert len(rows) == n_pad
    assert np.array_equal(np.sort(rows[rows >= 0]), np.arange(n)) and (rows[n:] == -1).all()
    nbr = rb.nbr_out.cpu().numpy()
    tab_rm = tp.rows.cpu().numpy()
    assert np.array_equal(tab_rm[:, :27], nbr.T) and (tab_rm[:, 27:] == -1).all()       # the row-major twin of the k-major table
    masks = ((nbr.T >= 0) * (1 << np.arange(27))).sum(1).astype(np.int64)
    assert np.array_equal(masks, tp.masks.cpu().numpy().astype(np.int64))
    mp = tp.masks_p.cpu().numpy().astype(np.int64)
    live = rows >= 0
    assert np.array_equal(mp[live], masks[rows[live]]) and (mp[~live] == 0).all()
    tab = np.zeros((n_pad, 29), np.int64)                                                 # position-ordered view used below: [27] = mask
    tab[:, 27] = mp
    nblk = (n + 1023) // 1024
    starts = [min((nblk * r // 8) * 1024, n) for r in range(9)]
    starts[8] = n
    for r in range(8):                                                                    # a row stays inside its region (= on its XCD)
        seg = rows[starts[r]:starts[r + 1] if r < 7 else n_pad]
        seg = seg[seg >= 0]
        assert ((seg >= starts[r]) & (seg < starts[r + 1])).all(), r
    tiles = tile_of.cpu().numpy().reshape(8, 512, -1)                                     # [region][wave of the region][slot]
    assert tiles.shape[2] % g == 0
    n_tiles = n_pad // 16
    cost = np.array([bin(int(np.bitwise_or.reduce(tab[16 * t:16 * t + 16, 27]))).count("1") for t in range(n_tiles)])
    seen = np.zeros(n_tiles, int)
    cu_work = []
    for r in range(8):
        t0, t1 = starts[r] // 16, (starts[r + 1] // 16 if r < 7 else n_tiles)
        w = tiles[r]
        assert ((w == -1) | ((w >= t0) & (w < t1))).all(), r
        filled = w >= 0
        assert (filled[:, :-1] >= filled[:, 1:]).all()                                      # slots are filled front to back
        np.add.at(seen, w[filled], 1)
        wave_work = np.where(filled, cost[np.maximum(w, 0)], 0).sum(1).reshape(128, 4)      # [workgroup][wave]
        if g == 4:
            # the tiles of a pass cost the same up to the steps of the sorted list (one tile from each of four rounds would spread by ~10)
            c = np.where(filled, cost[np.maximum(w, 0)], -1).reshape(512, -1, g)
            full = (c >= 0).all(2)
            spread = (c.max(2) - c.min(2))[full]
            assert len(spread) and np.percentile(spread, 90) <= 2, np.percentile(spread, [50, 90, 100])
        if t1 - t0 >= 512:
            assert np.percentile(wave_work.max(1) - wave_work.min(1), 90) <= 4             # the 4 waves of a workgroup carry near-equal work (quads of the sorted list)
        cu_work += list(wave_work.sum(1).reshape(4, 32).sum(0))                            # workgroups j, j+32, j+64, j+96 share a CU
    assert (seen == 1).all()                                                                # every tile exactly once
    cu_work = np.array(cu_work, float).reshape(8, 32)
    per_region = cu_work.mean(1, keepdims=True)
    # CU bins of an XCD: one quad per round each, so the bound is the spread of the sorted list's head (see k_plan_deal): <= 1.5x here (7 rounds),
    # 1.04-1.09x at the bench's 16-34 rounds (tools/conv_trace.py); units of four quads (g = 4, four rounds): <= 2x
    assert (cu_work <= (2.0 if g == 4 else 1.5) * per_region + 27 * (4 if g == 4 else 1)).all(), (cu_work.max(1) / per_region[:, 0])
    useful = sum(bin(int(m)).count("1") for m in tab[:, 27])
    assert useful / (16.0 * cost.sum()) >= 0.75                                           # useful / executed MFMA steps (consecutive rows: ~0.55)
    if fused:
        # a table has exactly ONE plan: rows keep their table order inside a class, tiles theirs inside a cost bucket (plan_region_body_stable) --
        # so the BatchNorm column sums the conv epilogue makes per workgroup, and everything behind them, are the same in every run
        classes = np.array([(((m >> 9) & 0x1ff) | (((1 if m & 0x1ff else 0) | (2 if (m >> 18) & 0x1ff else 0)) << 9)) if (m >> 9) & 0x1ff
                            else (2048 | (((1 if m & 0x1ff else 0) | (2 if (m >> 18) & 0x1ff else 0)) << 9) | ((m & 0x1ff) or ((m >> 18) & 0x1ff))) for m in masks])
        for r in range(8):
            seg = rows[starts[r]:starts[r + 1]]
            seg = seg[seg >= 0]
            key = classes[seg]
            assert (np.diff(key) >= 0).all(), r                                            # class-major
            same = np.diff(key) == 0
            assert (np.diff(seg)[same] > 0).all(), r                                       # table order inside a class
        for _ in range(2):
            tp2 = Fsp.TablePlan(rb.nbr_out, n, rb.K, rb.rows_out, rb.masks_out, g=g)
            assert torch.equal(tp2.perm, tp.perm) and torch.equal(tp2.masks_p, tp.masks_p) and torch.equal(tp2.tiles(g), tile_of)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(3, 16), (4, 16), (16, 16), (16, 32), (32, 64), (64, 64), (64, 128), (128, 128), (5, 7)])
def test_hip_conv_forward_backward_vs_oracle(cuda, hip_lib, cin, cout):
    import seevcn_amd.spconv as spconv
    rng = np.random.default_rng(5)
    batch, shape = 2, (9, 48, 40)
    coords = _rand_coords(rng, 2500, batch, shape)
    feats = rng.normal(size=(len(coords), cin)).astype(np.float32)
    for subm in (True, False):
        torch.manual_seed(0)
        conv = (spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="k") if subm
                else spconv.SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=False)).to(cuda)
        x = spconv.SparseConvTensor(torch.from_numpy(feats).to(cuda).requires_grad_(True), torch.from_numpy(coords).to(cuda), shape, batch)
        y = conv(x)
        w = osp.weight_to_kio(conv.weight.detach().cpu().numpy())
        if subm:
            nbr = osp.rulebook_subm(coords, shape, 3)
            oc = coords
        else:
            oc, nbr, _, _ = osp.rulebook_sparse(coords, shape, 3, 2, 1)
        ref = osp.conv_forward(feats, nbr, w)
        assert np.array_equal(y.indices.cpu().numpy(), oc)
        assert _ok(y.features.detach().cpu().numpy(), ref, name='forward')
        go = rng.normal(size=ref.shape).astype(np.float32)
        y.features.backward(torch.from_numpy(go).to(cuda))
        gf, gw = osp.conv_backward(feats, nbr, w, go)
        assert _ok(x.features.grad.cpu().numpy(), gf, name='data gradient')
        gw_hip = osp.weight_to_kio(conv.weight.grad.cpu().numpy())
        assert _ok(gw_hip, gw, atol_frac=5e-4, name='weight gradient')      # sums over ~2500 rows in fp32


@pytest.mark.gpu
def test_hip_conv_is_bitwise_reproducible(cuda, hip_lib):
    import seevcn_amd.spconv as spconv
    rng = np.random.default_rng(6)
    coords = _rand_coords(rng, 4000, 2, (9, 64, 64))
    feats = torch.from_numpy(rng.normal(size=(len(coords), 32)).astype(np.float32)).to(cuda)
    conv = spconv.SubMConv3d(32, 32, 3, padding=1, bias=False).to(cuda)
    outs = []
    for _ in range(2):
        x = spconv.SparseConvTensor(feats.clone().requires_grad_(True), torch.from_numpy(coords).to(cuda), (9, 64, 64), 2)
        y = conv(x)
        conv.weight.grad = None
        y.features.sum().backward()
        outs.append((y.features.detach().clone(), x.features.grad.clone(), conv.weight.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_hip_dense_and_height_compression(cuda, hip_lib):
    import seevcn_amd.spconv as spconv
    from seevcn_amd.pcdet.models.backbones_2d import map_to_bev
    rng = np.random.default_rng(7)
    # (2,50,44): cells per scene not a multiple of 64 -> 16-byte-store kernel; (2,32,24) -> the transposing 64-cell kernel, sparse and
    # nearly full; 6 channels -> scalar kernel; 300 channels -> beyond the LDS tile of the transposing kernels
    for batch, shape, c, n in [(3, (2, 50, 44), 128, 1500), (2, (2, 32, 24), 128, 400), (1, (2, 32, 24), 64, 1500), (2, (1, 16, 16), 6, 40),
                               (1, (2, 32, 24), 300, 200)]:
        coords = _rand_coords(rng, n, batch, shape, clustered=False)
        feats = rng.normal(size=(len(coords), c)).astype(np.float32)
        f = torch.from_numpy(feats).to(cuda).requires_grad_(True)
        t = spconv.SparseConvTensor(f, torch.from_numpy(coords).to(cuda), shape, batch)
        hc = map_to_bev.__all__["HeightCompression"]({"NUM_BEV_FEATURES": c * shape[0]})
        bd = hc({"encoded_spconv_tensor": t, "encoded_spconv_tensor_stride": 8})
        ref = osp.dense(feats, coords, batch, shape).reshape(batch, c * shape[0], shape[1], shape[2])
        assert bd["spatial_features"].shape == ref.shape and bd["spatial_features_stride"] == 8
        assert np.array_equal(bd["spatial_features"].detach().cpu().numpy(), ref)
        g = rng.normal(size=ref.shape).astype(np.float32)
        bd["spatial_features"].backward(torch.from_numpy(g).to(cuda))
        gd = g.reshape(batch, c, *shape)
        assert np.array_equal(f.grad.cpu().numpy(), gd[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]])


@pytest.mark.gpu
def test_hip_height_compression_in_channels_last_memory(cuda, hip_lib):
    """HeightCompression in front of a channels_last BaseBEVBackbone (feeds_bev_backbone, >= 8 scenes): spatial_features is WRITTEN with channels_last
    strides (sv_sparse_to_dense_nhwc) -- the same (N, C*D, H, W) values as the oracle's dense(), a tensor BaseBEVBackbone's
    .contiguous(memory_format=channels_last) returns as it is -- and its backward takes the gradient in either layout; shapes the kernel does not take
    (H W % 16 != 0, 6 channels) fall back to the contiguous view."""
    import seevcn_amd.spconv as spconv
    from seevcn_amd.pcdet.models.backbones_2d import map_to_bev
    rng = np.random.default_rng(11)
    for batch, shape, c, n, takes in [(8, (2, 32, 24), 128, 1500, True), (9, (2, 20, 16), 64, 900, True), (8, (1, 16, 16), 32, 300, True),
                                      (8, (5, 16, 8), 16, 700, True), (8, (2, 25, 10), 64, 400, False), (8, (2, 16, 16), 6, 100, False)]:
        coords = _rand_coords(rng, n, batch, shape, clustered=False)
        feats = rng.normal(size=(len(coords), c)).astype(np.float32)
        f = torch.from_numpy(feats).to(cuda).requires_grad_(True)
        t = spconv.SparseConvTensor(f, torch.from_numpy(coords).to(cuda), shape, batch)
        hc = map_to_bev.__all__["HeightCompression"]({"NUM_BEV_FEATURES": c * shape[0]})
        hc.feeds_bev_backbone = True
        sf = hc({"encoded_spconv_tensor": t, "encoded_spconv_tensor_stride": 8})["spatial_features"]
        ref = osp.dense(feats, coords, batch, shape).reshape(batch, c * shape[0], shape[1], shape[2])
        assert sf.shape == ref.shape and np.array_equal(sf.detach().cpu().numpy(), ref)
        assert sf.is_contiguous(memory_format=torch.channels_last) == (takes or c * shape[0] == 1)
        if takes:
            assert sf.contiguous(memory_format=torch.channels_last).data_ptr() == sf.data_ptr()       # no copy in front of the NHWC convolutions
        gd_full = rng.normal(size=ref.shape).astype(np.float32)
        want = gd_full.reshape(batch, c, *shape)[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]]
        for fmt in (torch.channels_last, torch.contiguous_format):                                      # the gradient in the backbone's layout, or in NCHW
            f.grad = None
            g = torch.from_numpy(gd_full).to(cuda).contiguous(memory_format=fmt)
            sf.backward(g, retain_graph=True)
            assert np.array_equal(f.grad.cpu().numpy(), want), (shape, c, fmt)
    # below 8 scenes, or without a BaseBEVBackbone behind it: the reference's contiguous view
    coords = _rand_coords(rng, 500, 4, (2, 32, 24), clustered=False)
    t = spconv.SparseConvTensor(torch.from_numpy(rng.normal(size=(len(coords), 64)).astype(np.float32)).to(cuda), torch.from_numpy(coords).to(cuda), (2, 32, 24), 4)
    hc = map_to_bev.__all__["HeightCompression"]({"NUM_BEV_FEATURES": 128})
    hc.feeds_bev_backbone = True
    assert hc({"encoded_spconv_tensor": t, "encoded_spconv_tensor_stride": 8})["spatial_features"].is_contiguous()
    t8 = spconv.SparseConvTensor(t.features, t.indices, (2, 32, 24), 8)
    hc.feeds_bev_backbone = False
    assert hc({"encoded_spconv_tensor": t8, "encoded_spconv_tensor_stride": 8})["spatial_features"].is_contiguous()


@pytest.mark.gpu
def test_hip_backbone8x_vs_oracle(cuda, hip_lib):
    """DynMeanVFE -> VoxelBackBone8x (eval) on a small KITTI-geometry batch against the oracle chain."""
    import seevcn_amd.synth as synth
    from seeding import seeded_state_dict
    from seevcn_amd.pcdet.models import backbones_3d
    from seevcn_amd.pcdet.models.backbones_3d import vfe
    pts, _ = synth.make_scene_batch(2, seed=2000, n_az=100)
    pc_range, vs, grid = [0, -40, -3, 70.4, 40, 1], [0.05, 0.05, 0.1], [1408, 1600, 40]
    bd = {"batch_size": 2, "points": torch.from_numpy(pts).to(cuda)}
    bd = vfe.__all__["DynMeanVFE"](model_cfg={}, num_point_features=3, voxel_size=vs, grid_size=grid, point_cloud_range=pc_range)(bd)
    m = backbones_3d.__all__["VoxelBackBone8x"]({}, 3, grid)
    sd = seeded_state_dict(m, seed=1)
    m.load_state_dict(sd)
    m = m.to(cuda).eval()
    with torch.no_grad():
        bd = m(bd)
    ref = osp.voxel_backbone8x_forward({k: v.numpy() for k, v in sd.items()}, bd["voxel_features"].cpu().numpy(),
                                       bd["voxel_coords"].cpu().numpy(), 2, m.sparse_shape)
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        t = bd["multi_scale_3d_features"][name]
        f, c, shape = ref[name]
        assert list(t.spatial_shape) == list(shape), name
        assert np.array_equal(t.indices.cpu().numpy(), c), name
        assert _ok(t.features.cpu().numpy(), f, name=name)
    t = bd["encoded_spconv_tensor"]
    f, c, shape = ref["out"]
    assert list(t.spatial_shape) == [2, 200, 176] == list(shape)
    assert np.array_equal(t.indices.cpu().numpy(), c) and _ok(t.features.cpu().numpy(), f, name='conv_out')


@pytest.mark.gpu
def test_hip_backbone8x_train_step_gradients_vs_oracle_chain(cuda, hip_lib):
    """The benchmarked step itself: DynMeanVFE -> VoxelBackBone8x (TRAIN mode: batch-statistics BatchNorm) -> HeightCompression -> loss ->
    backward on 2 scenes.  Forward features, the input gradient and all 12 conv weight gradients + 24 BatchNorm parameter gradients against
    the float64 oracle chain (oracle/spconv_train.py), element-wise per channel (tests/tolerances.py)."""
    import seevcn_amd.synth as synth
    from oracle import spconv_train as ost
    from seeding import seeded_state_dict
    from tolerances import assert_close_per_channel
    from seevcn_amd.pcdet.models import backbones_3d
    from seevcn_amd.pcdet.models.backbones_2d import map_to_bev
    from seevcn_amd.pcdet.models.backbones_3d import vfe
    pts, _ = synth.make_scene_batch(2, seed=2000, n_az=90)
    pc_range, vs, grid = [0, -40, -3, 70.4, 40, 1], [0.05, 0.05, 0.1], [1408, 1600, 40]
    bd = {"batch_size": 2, "points": torch.from_numpy(pts).to(cuda)}
    bd = vfe.__all__["DynMeanVFE"](model_cfg={}, num_point_features=3, voxel_size=vs, grid_size=grid, point_cloud_range=pc_range)(bd)
    m = backbones_3d.__all__["VoxelBackBone8x"]({}, 3, grid)
    sd = seeded_state_dict(m, seed=1)
    m.load_state_dict(sd)
    m = m.to(cuda).train()
    feats = bd["voxel_features"].detach().clone().requires_grad_(True)
    bd["voxel_features"] = feats
    bd = map_to_bev.__all__["HeightCompression"]({"NUM_BEV_FEATURES": 256})(m(bd))
    dense = bd["spatial_features"]
    # loss = sum(dense * G) with a fixed random G: every output element gets its own upstream gradient
    G = torch.from_numpy(np.random.default_rng(5).normal(size=tuple(dense.shape)).astype(np.float32))
    (dense * G.to(cuda)).sum().backward()

    ref_dense, leaves, (ref_out, ref_coords, ref_shape) = ost.backbone8x_train_chain({k: v.numpy() for k, v in sd.items()}, feats.detach().cpu().numpy(),
                                                                                    bd["voxel_coords"].cpu().numpy(), 2, m.sparse_shape)
    (ref_dense * G.double()).sum().backward()
    t = bd["encoded_spconv_tensor"]
    assert np.array_equal(t.indices.cpu().numpy(), ref_coords) and list(t.spatial_shape) == list(ref_shape)
    assert_close_per_channel(t.features.detach().cpu().numpy(), ref_out.detach().numpy(), name="conv_out features (train-mode BN)")
    assert_close_per_channel(feats.grad.cpu().numpy(), leaves["input"].grad.numpy(), rtol=2e-3, atol_frac=2e-4, name="d loss / d voxel_features")
    checked = 0
    for key, p in m.named_parameters():
        want = leaves[key].grad.numpy()
        got = p.grad.detach().cpu().numpy()
        if got.ndim == 5:                                                  # (C_out, kz, ky, kx, C_in) -> (K, C_in, C_out)
            got = osp.weight_to_kio(got)
        # gradients are sums over 10^3..10^5 rows of products of O(1) terms in fp32: 2e-3 of each output channel's own largest value
        assert_close_per_channel(got, want, rtol=2e-3, atol_frac=2e-3, name="grad " + key)
        checked += 1
    assert checked == 12 + 24
    # running statistics moved like torch's BatchNorm1d (momentum 0.01, unbiased variance)
    assert int(m.conv_out[1].num_batches_tracked) == 1 and float((m.conv_out[1].running_mean - sd["conv_out.1.running_mean"].to(cuda)).abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,relu", [(134580, 64, True), (50001, 16, True), (7, 128, False), (2, 32, True), (300000, 32, False)])
def test_hip_fused_batchnorm_relu_matches_torch(cuda, hip_lib, n, c, relu):
    """sv_batchnorm_relu_forward/backward vs torch.nn.BatchNorm1d(eps=1e-3, momentum=0.01) [+ ReLU] (spconv_backbone.py:73), the
    reference's own norm layer: outputs, running statistics, input / weight / bias gradients; then eval mode."""
    import torch.nn as nn
    from seevcn_amd.spconv import norm
    g = torch.Generator().manual_seed(n + c)
    x0 = (torch.randn(n, c, generator=g) * 1.7 + 0.6).to(cuda)
    dy = torch.randn(n, c, generator=g).to(cuda)
    ref, mine = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(cuda), nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(cuda)
    with torch.no_grad():
        ref.weight.copy_(torch.rand(c, generator=g) + 0.5), ref.bias.copy_(torch.randn(c, generator=g) * 0.3)
    mine.load_state_dict(ref.state_dict())
    xr, xm = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    assert norm.fusable(mine, xm)
    for _ in range(2):                                        # two steps: running statistics accumulate
        yr = ref(xr)
        yr = torch.relu(yr) if relu else yr
        ym = norm.batch_norm_relu(mine, xm, relu)
    tol = dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ym, yr, **tol)
    torch.testing.assert_close(mine.running_mean, ref.running_mean, **tol)
    torch.testing.assert_close(mine.running_var, ref.running_var, **tol)
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 2
    yr.backward(dy), ym.backward(dy)
    scale = float(xr.grad.abs().max())
    assert float((xm.grad - xr.grad).abs().max()) <= 1e-3 * scale + 1e-6
    torch.testing.assert_close(mine.weight.grad, ref.weight.grad, rtol=1e-3, atol=1e-3 * float(ref.weight.grad.abs().max()))
    torch.testing.assert_close(mine.bias.grad, ref.bias.grad, rtol=1e-3, atol=1e-3 * float(ref.bias.grad.abs().max()))
    ref.eval(), mine.eval()
    with torch.no_grad():
        ye = ref(x0)
        ye = torch.relu(ye) if relu else ye
        torch.testing.assert_close(norm.batch_norm_relu(mine, x0, relu), ye, **tol)


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["kitti16", "nuscenes_stress"])
def test_hip_sparse_conv_full_size_properties(cuda, hip_lib, config):
    """BASELINE sizes (config 3: 16 KITTI scenes, ~213 k voxels; config 5: 300 k points / scene on the 1440 x 1440 x 40 grid) through
    size-independent properties: rulebook symmetry / inverse tables / sorted unique outputs, linearity of the conv, and the adjoint
    identities <conv(X; W), dY> = <X, bwd_data(dY; W)> = <W, wgrad(X, dY)> that tie forward, data gradient and weight gradient."""
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet.ops import voxel_ops
    from seevcn_amd.spconv import functional as Fsp
    if config == "kitti16":
        bs = 16
        pts, _ = synth.make_scene_batch(bs, seed=2000)
        pc_range, vs, grid = [0, -40, -3, 70.4, 40, 1], [0.05, 0.05, 0.1], [1408, 1600, 40]
    else:
        bs = 3
        pts, _ = synth.make_scene_batch(bs, seed=4000, n_beams=32, elev=(-30.0, 10.0), az=(-180.0, 180.0), n_az=940, n_sweeps=10,
                                        box_area=((-50.0, 50.0), (-50.0, 50.0)), max_range=54.0, z_shift=0.0)
        pc_range, vs, grid = [-54, -54, -5, 54, 54, 3], [0.075, 0.075, 0.2], [1440, 1440, 40]
    p = torch.from_numpy(pts).to(cuda)
    feats, coords, _ = voxel_ops.voxelize_dynamic(p, pc_range, vs, grid, bs)
    shape = [grid[2] + 1, grid[1], grid[0]]
    n = coords.shape[0]
    assert n > (150000 if config == "kitti16" else 100000)
    gen = torch.Generator(device=cuda).manual_seed(0)
    # --- submanifold rulebook: centre = identity, table symmetric under offset reversal
    rb = Fsp.build_subm_rulebook(coords, bs, shape, [3, 3, 3])
    nbr = rb.nbr_out
    rows = torch.arange(n, device=cuda, dtype=torch.int32)
    assert torch.equal(nbr[13], rows)
    for k in (0, 5, 12):
        j = nbr[k].long()
        ok = j >= 0
        assert torch.equal(nbr[26 - k][j[ok]], rows[ok])
    # --- strided rulebook: outputs sorted by key and unique, the two tables are inverses of each other
    rs = Fsp.build_sparse_rulebook(coords, bs, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    oc = rs.out_indices.long()
    key = ((oc[:, 0] * rs.out_shape[0] + oc[:, 1]) * rs.out_shape[1] + oc[:, 2]) * rs.out_shape[2] + oc[:, 3]
    assert bool((key[1:] > key[:-1]).all())
    assert int((rs.nbr_out >= 0).sum()) == int((rs.nbr_in >= 0).sum())
    for k in (0, 13, 26):
        o = rs.nbr_in[k].long()
        ok = o >= 0
        assert torch.equal(rs.nbr_out[k][o[ok]], rows[ok])
    # --- conv: linearity and adjointness on both rulebooks (64 -> 64 channels: the dominant kernel)
    for book, n_in, n_out in ((rb, n, n), (rs, n, rs.n_out)):
        cin = cout = 64
        x1, x2 = (torch.randn((n_in, cin), device=cuda, generator=gen) for _ in range(2))
        w = torch.randn((book.K, cin, cout), device=cuda, generator=gen) * 0.1
        dy = torch.randn((n_out, cout), device=cuda, generator=gen)
        pf = book.plan("fwd", cin, cout)
        conv = lambda x: Fsp.gather_gemm_planned(x, pf, Fsp.fragment_cache.get(w)[0], n_out, book.K, cin, cout)
        y1, y2, y12 = conv(x1), conv(x2), conv(2.0 * x1 - 3.0 * x2)
        scale = float(y12.abs().max())
        assert float((y12 - (2.0 * y1 - 3.0 * y2)).abs().max()) <= 1e-4 * scale
        dx = Fsp.gather_gemm_planned(dy, book.plan("bwd", cout, cin), Fsp.fragment_cache.get(w)[1], n_in, book.K, cout, cin)
        dw = Fsp.wgrad(x1, book.nbr_out, dy, book.K, cin, cout)
        lhs = float((y1.double() * dy.double()).sum())
        assert abs(lhs - float((x1.double() * dx.double()).sum())) <= 1e-4 * abs(lhs) + 1e-3
        assert abs(lhs - float((w.double() * dw.double()).sum())) <= 1e-4 * abs(lhs) + 1e-3
    # --- dense scatter / gather round trip on the backbone's last grid
    last = Fsp.build_sparse_rulebook(coords, bs, shape, [3, 3, 3], [8, 8, 8], [1, 1, 1])
    f = torch.randn((last.n_out, 128), device=cuda, generator=gen)
    import seevcn_amd.spconv as spconv
    t = spconv.SparseConvTensor(f.clone().requires_grad_(True), last.out_indices, last.out_shape, bs)
    d = t.dense()
    assert int((d != 0).sum()) == int((f != 0).sum())
    d.backward(d.detach())
    assert torch.equal(t.features.grad, f)




@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 64), (64, 64), (64, 128)])
def test_hip_batchnorm_statistics_from_the_conv_epilogue(cuda, hip_lib, cin, cout):
    """conv -> BatchNorm1d(train) -> ReLU as one node: the statistics' first pass made in the planned conv kernel's epilogue (per-workgroup column
    sums next to the whole-row stores) against the separate reduction pass over the conv output -- outputs, running statistics, gradients."""
    import seevcn_amd.spconv as spconv
    from seevcn_amd.spconv import norm
    from tolerances import assert_close_per_channel
    rng = np.random.default_rng(21)
    batch, shape = 2, (21, 200, 176)
    coords = _rand_coords(rng, 60000, batch, shape)
    feats = rng.normal(size=(len(coords), cin)).astype(np.float32)
    res = []
    for flag in (True, False):
        torch.manual_seed(3)
        seq = spconv.SparseSequential(spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="s"), torch.nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01),
                                      torch.nn.ReLU()).to(cuda).train()
        x = torch.from_numpy(feats).to(cuda).requires_grad_(True)
        saved, norm.STATS_IN_CONV = norm.STATS_IN_CONV, flag
        try:
            y = seq(spconv.SparseConvTensor(x, torch.from_numpy(coords).to(cuda), list(shape), batch)).features
            (y * y).sum().backward()
        finally:
            norm.STATS_IN_CONV = saved
        res.append((y.detach().cpu().numpy(), seq[1].running_mean.cpu().numpy(), seq[1].running_var.cpu().numpy(), x.grad.cpu().numpy(),
                    seq[0].weight.grad.reshape(cout, -1).cpu().numpy(), int(seq[1].num_batches_tracked)))
    a, b = res
    assert a[5] == b[5] == 1
    assert_close_per_channel(a[0], b[0], rtol=1e-4, atol_frac=1e-5, name="output")
    np.testing.assert_allclose(a[1], b[1], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(a[2], b[2], rtol=1e-5, atol=1e-7)
    assert_close_per_channel(a[3], b[3], rtol=1e-3, atol_frac=1e-4, name="input gradient")
    assert_close_per_channel(a[4], b[4], rtol=1e-3, atol_frac=1e-4, name="weight gradient")


@pytest.mark.gpu
def test_hip_backbone_chain_equals_the_module_path(cuda, hip_lib):
    """VoxelBackBone8x in training mode through the launch-list chain (one autograd node, sv_run_ops) against the per-module path
    (SparseSequential: one node per block): the same kernels with the same arguments, so outputs, running statistics and every gradient are
    bit-identical; a gradient that enters at a multi-scale tap (x_conv3, as PV-RCNN's set abstraction sends it) is carried as well.
    Both in the DEFAULT configuration (BatchNorm statistics summed in the conv epilogues, in the order of the plan's tiles -- the two networks
    build their own rulebooks and plans, and a table has exactly one plan) and with the statistics made by their own reduction pass; and with the
    chain's BatchNorm + ReLU applied by the consumers as they gather (chain.BN_FOLD, the default: no normalised tensor is written between the blocks,
    x_conv3 is made when it is read) as well as with every block writing its output -- the module path always writes them: bit-identical either way."""
    import copy
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet.models import backbones_3d
    from seevcn_amd.pcdet.ops import voxel_ops
    from seevcn_amd.spconv import chain, norm
    pts, _ = synth.make_scene_batch(2, seed=2000, n_az=120)
    g = dict(point_cloud_range=[0, -40, -3, 70.4, 40, 1], voxel_size=[0.05, 0.05, 0.1], grid_size=[1408, 1600, 40])
    f, c, _ = voxel_ops.voxelize_dynamic(torch.from_numpy(pts).to(cuda), g["point_cloud_range"], g["voxel_size"], g["grid_size"], 2)
    torch.manual_seed(0)
    net1 = backbones_3d.__all__['VoxelBackBone8x']({}, 3, g['grid_size']).to(cuda).train()
    for stats_in_conv, fold in ((True, True), (False, True), (True, False)):
        net2 = copy.deepcopy(net1)
        net3 = copy.deepcopy(net1)
        res = []
        for net, off in ((net2, False), (net3, True)):
            saved, chain.CHAIN_OFF = chain.CHAIN_OFF, off
            saved_stats, norm.STATS_IN_CONV = norm.STATS_IN_CONV, stats_in_conv
            saved_fold, chain.BN_FOLD = chain.BN_FOLD, fold
            saved_bwd, chain.BWD_SUMS_IN_CONV = chain.BWD_SUMS_IN_CONV, False      # bit-identity: the BatchNorm backward sums in passes of their own, like the modules
            try:
                assert (net._chain_blocks() is not None)
                bd = net({'batch_size': 2, 'voxel_features': f.clone(), 'voxel_coords': c.clone()})
                x3t = bd['multi_scale_3d_features']['x_conv3']
                assert isinstance(x3t, chain.LazyTap) == (fold and not off)
                out, x3 = bd['encoded_spconv_tensor'].features, x3t.features
                assert (type(out.grad_fn).__name__ == "SparseChainFunctionBackward") == (not off)
                (out.square().sum() + (x3 * 0.5).sum()).backward()
            finally:
                chain.CHAIN_OFF = saved
                norm.STATS_IN_CONV = saved_stats
                chain.BN_FOLD = saved_fold
                chain.BWD_SUMS_IN_CONV = saved_bwd
            res.append((out.detach(), x3.detach(), [p.grad.clone() for p in net.parameters()], [b.clone() for b in net.buffers()]))
        a, b = res
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (stats_in_conv, fold)
        for (n1, _), g1, g2 in zip(net1.named_parameters(), a[2], b[2]):
            assert torch.equal(g1, g2), (n1, stats_in_conv, fold)
        for (n1, _), b1, b2 in zip(net1.named_buffers(), a[3], b[3]):
            assert torch.equal(b1, b2), (n1, stats_in_conv, fold)


@pytest.mark.gpu
def test_hip_chain_with_statistics_in_the_conv_epilogues_matches_the_separate_passes(cuda, hip_lib):
    """Default chain: BatchNorm's forward statistics come out of the conv epilogue and its BACKWARD sums out of the epilogue of the data-gradient
    launch above (sv_sparse_conv_dgrad_planned_bn + sv_batchnorm_relu_backward_partial).  Same numbers as the chain with every BatchNorm making
    its sums in passes of its own, up to the summation order (plan order instead of row order): 1e-5 of each tensor's largest entry."""
    import copy
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet.models import backbones_3d
    from seevcn_amd.pcdet.ops import voxel_ops
    from seevcn_amd.spconv import chain, norm
    pts, _ = synth.make_scene_batch(2, seed=2001, n_az=120)
    g = dict(point_cloud_range=[0, -40, -3, 70.4, 40, 1], voxel_size=[0.05, 0.05, 0.1], grid_size=[1408, 1600, 40])
    f, c, _ = voxel_ops.voxelize_dynamic(torch.from_numpy(pts).to(cuda), g["point_cloud_range"], g["voxel_size"], g["grid_size"], 2)
    torch.manual_seed(1)
    net1 = backbones_3d.__all__['VoxelBackBone8x']({}, 3, g['grid_size']).to(cuda).train()
    with torch.no_grad():                                                  # negative and positive gammas, non-zero betas: both ReLU branches matter
        for m in net1.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(-1.5, 1.5), m.bias.uniform_(-0.5, 0.5)
    net2, net3 = copy.deepcopy(net1), copy.deepcopy(net1)
    res, indice_dict = [], None
    # run 1: default.  run 2: forward statistics in the conv epilogue, backward sums in passes of their own -- on the same plans the forward is
    # bit-identical (same ReLU branches), so the gradients differ by the summation order only.  run 3: nothing in the conv epilogues.
    for net, in_conv, bwd_in_conv in ((net1, True, True), (net2, True, False), (net3, False, False)):
        saved, norm.STATS_IN_CONV = norm.STATS_IN_CONV, in_conv
        saved_off, chain.CHAIN_OFF = chain.CHAIN_OFF, False
        saved_bwd, chain.BWD_SUMS_IN_CONV = chain.BWD_SUMS_IN_CONV, bwd_in_conv
        try:
            bd = {'batch_size': 2, 'voxel_features': f.clone(), 'voxel_coords': c.clone()}
            if indice_dict is not None:
                bd['spconv_indice_dict'] = indice_dict                    # same rulebooks and plans for all runs
            bd = net(bd)
            indice_dict = bd['encoded_spconv_tensor'].indice_dict
            out, x3 = bd['encoded_spconv_tensor'].features, bd['multi_scale_3d_features']['x_conv3'].features
            (out.square().sum() + (x3 * 0.5).sum()).backward()           # the x_conv3 tap carries an outside gradient: that block keeps its own pass
        finally:
            norm.STATS_IN_CONV, chain.CHAIN_OFF, chain.BWD_SUMS_IN_CONV = saved, saved_off, saved_bwd
        res.append((out.detach(), [p.grad.clone() for p in net.parameters()]))
    (o1, g1), (o2, g2), (o3, g3) = res
    assert torch.equal(o1, o2)
    for (name, _), a, b in zip(net1.named_parameters(), g1, g2):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7, (name, float((a - b).abs().max()), float(b.abs().max()))
    # against the fully separate passes a handful of the 10^6 activations may take the other ReLU branch (statistics equal to ~1e-7 only)
    assert float((o1 - o3).abs().max()) <= 1e-5 * float(o3.abs().max())
    for (name, _), a, b in zip(net1.named_parameters(), g1, g3):
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-7, (name, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.gpu
def test_hip_launch_list_executor_runs_in_order_and_reports_errors(cuda, hip_lib):
    """sv_run_ops: a list of BatchNorm forward rows equals the same calls made one by one (same kernels, same arguments); an unknown code and a
    failing operation stop the list with the library's error text; an empty list is fine."""
    from seevcn_amd import _lib
    from seevcn_amd.spconv import chain, norm
    g = torch.Generator().manual_seed(0)
    n, c = 5000, 32
    x = torch.randn(n, c, generator=g).to(cuda)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(cuda), torch.randn(c, generator=g).to(cuda)

    def stats():
        return torch.zeros(c, device=cuda), torch.ones(c, device=cuda), torch.zeros((), dtype=torch.int64, device=cuda)

    rm1, rv1, nb1 = stats()
    y1, mean1, istd1 = norm.bn_forward_raw(x, gamma, beta, rm1, rv1, 0.01, 1e-3, True, True, nb1)
    z1, _, _ = norm.bn_forward_raw(y1, gamma, beta, rm1, rv1, 0.01, 1e-3, True, False, nb1)
    rm2, rv2, nb2 = stats()
    y2, z2 = torch.empty_like(x), torch.empty_like(x)
    m2, i2, m3, i3 = (torch.empty(c, device=cuda) for _ in range(4))
    scratch = norm._scratch(c, cuda)
    me = (chain._bits(0.01), chain._bits(1e-3))
    rows = [chain._row(chain.OP_BN_FWD, i=(c, 1, 1, 0), n=(n,), f=me, p=(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm2.data_ptr(), rv2.data_ptr(),
                                                                     scratch.data_ptr(), y2.data_ptr(), m2.data_ptr(), i2.data_ptr(), nb2.data_ptr())),
            chain._row(chain.OP_BN_FWD, i=(c, 1, 0, 0), n=(n,), f=me, p=(y2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm2.data_ptr(), rv2.data_ptr(),
                                                                     scratch.data_ptr(), z2.data_ptr(), m3.data_ptr(), i3.data_ptr(), nb2.data_ptr()))]
    chain._run(rows, "two BatchNorm rows")
    assert torch.equal(y1, y2) and torch.equal(z1, z2) and torch.equal(mean1, m2) and torch.equal(istd1, i2)
    assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2) and int(nb2) == 2 == int(nb1)
    chain._run([], "empty list")
    with pytest.raises(_lib.SeevcnHipError, match="unknown operation 99 at position 1"):
        chain._run([rows[0], chain._row(99)], "bad code")
    with pytest.raises(_lib.SeevcnHipError, match="null pointer"):
        chain._run([chain._row(chain.OP_BN_FWD, i=(c, 1, 1, 0), n=(n,), f=me)], "null pointers")


def _same_rulebook(a, b, tag):
    assert a.subm == b.subm and a.n_in == b.n_in and a.n_out == b.n_out and list(a.out_shape) == list(b.out_shape) and a.ksize == b.ksize, tag
    for name in ("out_indices", "nbr_in", "nbr_out", "rows_in", "rows_out", "masks_in", "masks_out"):
        x, y = getattr(a, name), getattr(b, name)
        assert (x is None) == (y is None), (tag, name)
        if x is not None:
            assert x.shape == y.shape and torch.equal(x, y), (tag, name)


@pytest.mark.gpu
@pytest.mark.parametrize("lazy_count", [False, True])
def test_hip_network_index_equals_the_layer_by_layer_build(cuda, hip_lib, lazy_count):
    """build_network_index (strided levels counted end to end on the device -- level l + 1 marks from level l's freshly written site list --
    ONE read for every count, then all tables through the levels' cell maps in three launches and all plans in one) == build_subm_rulebook /
    build_sparse_rulebook layer after layer: coordinates, both tables, row-major twins and masks, bit for bit; the plans are valid plans of the
    same tables; the persistent indices and cell maps are left all-zero (a second build on them gives the same tables).  lazy_count: the
    coordinate tensor has capacity rows (garbage behind the first n0) and n0 lives on the device."""
    from seevcn_amd import _lib
    from seevcn_amd.spconv import functional as Fsp
    rng = np.random.default_rng(11)
    batch, shape = 3, (21, 96, 80)
    coords = torch.from_numpy(_rand_coords(rng, 20000, batch, shape)).to(cuda)
    n0 = coords.shape[0]
    S = Fsp.ConvSpec
    specs = [S("subm1", True, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 16, 16), S("subm1", True, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 16, 16),
             S("sp2", False, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), 16, 32), S("subm2", True, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 32, 32),
             S("sp3", False, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), 32, 64), S("subm3", True, (3, 1, 3), (1, 1, 1), (1, 0, 1), (1, 1, 1), 64, 64),
             S("sp4", False, (3, 3, 3), (2, 2, 2), (0, 1, 1), (1, 1, 1), 64, 64), S("subm4", True, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 64, 64),
             S("down", False, (3, 1, 1), (2, 1, 1), (0, 0, 0), (1, 1, 1), 64, 128)]
    want, idx, sh = {}, coords, list(shape)
    for sp in specs:
        if sp.key in want:
            continue
        if sp.subm:
            want[sp.key] = Fsp.build_subm_rulebook(idx, batch, sh, sp.ksize, sp.dilation)
        else:
            want[sp.key] = rb = Fsp.build_sparse_rulebook(idx, batch, sh, sp.ksize, sp.stride, sp.padding, sp.dilation)
            idx, sh = rb.out_indices, list(rb.out_shape)
    torch.cuda.synchronize()
    _lib.workspace.reset()            # fresh (zeroed) workspaces: the layer-by-layer builders leave their scan output (chunk bases) behind
    if lazy_count:
        given = torch.cat([coords, torch.full((777, 4), 1 << 20, dtype=torch.int32, device=cuda)])      # capacity rows behind n0: never read
        n0_dev = torch.tensor([n0], dtype=torch.int32, device=cuda)
    else:
        given, n0_dev = coords, None
    for rep in range(2):
        got_n0, got = Fsp.build_network_index(given, batch, shape, specs, n0_dev=n0_dev)
        assert got_n0 == n0 and sorted(got) == sorted(want)
        for key in want:
            _same_rulebook(got[key], want[key], (rep, key))
        # level identities: a strided table's output sites ARE the next level's input sites (get_rulebook compares identities)
        assert got["sp3"].in_indices is got["sp2"].out_indices and got["subm2"].out_indices is got["sp2"].out_indices
        # plans: a permutation of the rows into regions, masks carried along, every tile dealt exactly once
        for key, rb in got.items():
            for pkey, tp in rb._plans.items():
                perm, n_rows = tp.perm.cpu().numpy(), tp.n_rows
                assert np.array_equal(np.sort(perm[perm >= 0]), np.arange(n_rows)), (key, pkey)
                assert np.array_equal(tp.masks_p.cpu().numpy()[perm >= 0], tp.masks.cpu().numpy()[perm[perm >= 0]]), (key, pkey)
                (g,) = tp._tiles_lazy                                   # the tiles-per-wave value the plan was dealt for
                t = tp.tiles(g).cpu().numpy()
                assert np.array_equal(np.sort(t[t >= 0]), np.arange((n_rows + 15) // 16)), (key, pkey)
    # nothing left behind in the persistent workspaces this build touched (bitmaps of the strided levels, cell maps of all levels)
    for (kind, name, *_), buf in _lib.workspace._bufs.items():
        if kind == "p" and (name.startswith("rb_index_") or name.startswith("rb_cellmap_")):
            assert int(buf.count_nonzero()) == 0, name
    # oracle for the first two strided levels
    oc, nbr_out, nbr_in, _ = osp.rulebook_sparse(coords.cpu().numpy(), shape, 3, 2, 1)
    assert np.array_equal(got["sp2"].out_indices.cpu().numpy(), oc) and np.array_equal(got["sp2"].nbr_out.cpu().numpy(), nbr_out)
    oc2, nbr_out2, _, _ = osp.rulebook_sparse(oc, got["sp2"].out_shape, 3, 2, 1)
    assert np.array_equal(got["sp3"].out_indices.cpu().numpy(), oc2) and np.array_equal(got["sp3"].nbr_out.cpu().numpy(), nbr_out2)


@pytest.mark.gpu
def test_hip_network_index_declines_what_it_does_not_take(cuda, hip_lib):
    """Wider kernels, an empty input or a cell map beyond the budget: None before anything is launched -- the caller builds layer by layer."""
    from seevcn_amd.spconv import functional as Fsp
    S = Fsp.ConvSpec
    coords = torch.from_numpy(_rand_coords(np.random.default_rng(3), 500, 2, (11, 40, 40))).to(cuda)
    assert Fsp.build_network_index(coords, 2, (11, 40, 40), [S("a", False, (5, 5, 5), (2, 2, 2), (2, 2, 2), (1, 1, 1), 16, 16)]) is None
    assert Fsp.build_network_index(coords[:0], 2, (11, 40, 40), [S("a", True, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 16, 16)]) is None
    saved, Fsp.CELLMAP_MAX_BYTES = Fsp.CELLMAP_MAX_BYTES, 1024
    try:
        assert Fsp.build_network_index(coords, 2, (11, 40, 40), [S("a", True, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 16, 16)]) is None
    finally:
        Fsp.CELLMAP_MAX_BYTES = saved


@pytest.mark.gpu
def test_hip_backbone_on_the_batched_index_equals_the_layer_by_layer_index(cuda, hip_lib):
    """VoxelBackBone8x forward + backward with its rulebooks and plans from build_network_index against the same network on rulebooks built
    layer by layer (SEEVCN_INDEX_BATCH=0): identical tables, so -- with the BatchNorm statistics in passes of their own, which takes the plan's
    row order out of the sums -- outputs and gradients are bit-identical.  The voxel count is read lazily in the first run (pipeline.front's way)."""
    import copy
    import seevcn_amd.synth as synth
    from seevcn_amd import spconv
    from seevcn_amd.pcdet.models import backbones_3d
    from seevcn_amd.pcdet.ops import voxel_ops
    from seevcn_amd.spconv import conv as sconv, norm
    pts, _ = synth.make_scene_batch(2, seed=2003, n_az=120)
    g = dict(point_cloud_range=[0, -40, -3, 70.4, 40, 1], voxel_size=[0.05, 0.05, 0.1], grid_size=[1408, 1600, 40])
    fcap, ccap, _, nvox = voxel_ops.voxelize_dynamic(torch.from_numpy(pts).to(cuda), g["point_cloud_range"], g["voxel_size"], g["grid_size"], 2, sync=False)
    torch.manual_seed(0)
    net1 = backbones_3d.__all__['VoxelBackBone8x']({}, 3, g['grid_size']).to(cuda).train()
    net2 = copy.deepcopy(net1)
    res = []
    saved_stats, norm.STATS_IN_CONV = norm.STATS_IN_CONV, False
    try:
        for net, batched in ((net1, True), (net2, False)):
            saved, sconv.BATCH_INDEX = sconv.BATCH_INDEX, batched
            try:
                sp = spconv.SparseConvTensor(fcap.clone(), ccap.clone(), net.sparse_shape, 2)
                spconv.prebuild_rulebooks(net, sp, with_backward=True, n0_dev=nvox)
                assert sp.indices.shape[0] == int(nvox) == sp.features.shape[0]
                bd = net({'batch_size': 2, 'voxel_features': sp.features, 'voxel_coords': sp.indices, 'spconv_indice_dict': sp.indice_dict})
                out, x3 = bd['encoded_spconv_tensor'].features, bd['multi_scale_3d_features']['x_conv3'].features
                (out.square().sum() + (x3 * 0.5).sum()).backward()
            finally:
                sconv.BATCH_INDEX = saved
            res.append((out.detach(), x3.detach(), [p.grad.clone() for p in net.parameters()], bd['encoded_spconv_tensor'].indices.clone(), sp.indice_dict))
    finally:
        norm.STATS_IN_CONV = saved_stats
    a, b = res
    assert torch.equal(a[3], b[3]) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for key in b[4]:
        _same_rulebook(a[4][key], b[4][key], key)
    for (n1, _), g1, g2 in zip(net1.named_parameters(), a[2], b[2]):
        assert torch.equal(g1, g2), n1


@pytest.mark.gpu
def test_hip_chain_stands_down_for_hooks_on_any_walked_module(cuda, hip_lib):
    """A forward hook on a STAGE (backbone_3d.conv2 -- the usual way to tap features), on a nested post_act_block or on a ReLU fires in training
    mode exactly as in eval mode: the launch-list chain bypasses __call__ of all of them, so it must stand down.  A BatchNorm with momentum=None
    (cumulative average) falls back too instead of raising when the chain is flattened."""
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet.models import backbones_3d
    from seevcn_amd.pcdet.ops import voxel_ops
    from seevcn_amd.spconv import chain
    pts, _ = synth.make_scene_batch(2, seed=2002, n_az=96)
    g = dict(point_cloud_range=[0, -40, -3, 70.4, 40, 1], voxel_size=[0.05, 0.05, 0.1], grid_size=[1408, 1600, 40])
    f, c, _ = voxel_ops.voxelize_dynamic(torch.from_numpy(pts).to(cuda), g["point_cloud_range"], g["voxel_size"], g["grid_size"], 2)
    torch.manual_seed(0)
    net = backbones_3d.__all__['VoxelBackBone8x']({}, 3, g['grid_size']).to(cuda).train()

    def run():
        return net({'batch_size': 2, 'voxel_features': f.clone(), 'voxel_coords': c.clone()})

    blocks = net._chain_blocks()
    assert blocks is not None and net.conv2 in blocks.walked and net.conv2[0] in blocks.walked and net.conv2[0][2] in blocks.walked
    for target in (net.conv2, net.conv3[1], net.conv1[0][2]):
        fired = []
        h = target.register_forward_hook(lambda m, i, o: fired.append(type(m).__name__))
        try:
            run()
        finally:
            h.remove()
        assert len(fired) == 1, (target, fired)
    # no hook left: the chain takes the forward again (one autograd node for the whole backbone)
    out = run()['encoded_spconv_tensor'].features
    assert type(out.grad_fn).__name__ == "SparseChainFunctionBackward"
    net.conv3[1][1].momentum = None
    out = run()['encoded_spconv_tensor'].features                  # no TypeError: module path, torch's cumulative moving average
    assert type(out.grad_fn).__name__ != "SparseChainFunctionBackward" and torch.isfinite(out).all()


@pytest.mark.gpu
def test_hip_deferred_weight_gradient_reduction_is_bitwise_the_per_layer_one(cuda, hip_lib):
    """The chain's backward list with ONE slab-reduction launch for all layers at its end (SV_OP_WGRAD_DEFERRED, the default) against a reduction launch
    behind every weight gradient (SV_OP_WGRAD): the same partial slabs summed in the same order -- every gradient bit-identical."""
    import copy
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet.models import backbones_3d
    from seevcn_amd.pcdet.ops import voxel_ops
    from seevcn_amd.spconv import chain
    pts, _ = synth.make_scene_batch(2, seed=2004, n_az=120)
    g = dict(point_cloud_range=[0, -40, -3, 70.4, 40, 1], voxel_size=[0.05, 0.05, 0.1], grid_size=[1408, 1600, 40])
    f, c, _ = voxel_ops.voxelize_dynamic(torch.from_numpy(pts).to(cuda), g["point_cloud_range"], g["voxel_size"], g["grid_size"], 2)
    torch.manual_seed(3)
    net1 = backbones_3d.__all__['VoxelBackBone8x']({}, 3, g['grid_size']).to(cuda).train()
    net2 = copy.deepcopy(net1)
    res = []
    for net, defer in ((net1, True), (net2, False)):
        saved, chain.DEFER_WGRAD_REDUCE = chain.DEFER_WGRAD_REDUCE, defer
        try:
            bd = net({'batch_size': 2, 'voxel_features': f.clone(), 'voxel_coords': c.clone()})
            out = bd['encoded_spconv_tensor'].features
            assert type(out.grad_fn).__name__ == "SparseChainFunctionBackward"
            out.square().sum().backward()
        finally:
            chain.DEFER_WGRAD_REDUCE = saved
        res.append({k: p.grad.clone() for k, p in net.named_parameters()})
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
    assert all(torch.isfinite(v).all() and float(v.abs().max()) > 0 for k, v in res[0].items() if k.endswith("0.weight"))


def _synthetic_lidar_table(n, K, cuda, seed=11):
    """(K, n) table with the spread between offsets a LiDAR rulebook has: every row at the centre, 70 % in the dz = 0 plane, 15 % out of plane."""
    g = torch.Generator(device=cuda).manual_seed(seed)
    density = torch.tensor([1.0 if k == K // 2 else (0.7 if K // 3 <= k < 2 * K // 3 else 0.15) for k in range(K)], device=cuda)
    keep = torch.rand((K, n), device=cuda, generator=g) < density[:, None]
    src = (torch.arange(n, device=cuda, dtype=torch.int32)[None, :] * 7 + torch.arange(K, device=cuda, dtype=torch.int32)[:, None] * 1013) % n
    return torch.where(keep, src, torch.full_like(src, -1)).contiguous()


@pytest.mark.gpu
@pytest.mark.parametrize("n,K,pieces", [(100_000, 27, 1024), (100_000, 27, 2048), (3_000, 27, 1024), (700, 3, 2048), (64, 27, 1024)])
def test_hip_weight_gradient_plan_cuts_the_table_into_equal_pieces(cuda, hip_lib, n, K, pieces):
    """sv_wgrad_plan_build: units in (row eighth, offset, unit) order; the cuts are monotone and cover every 64-row unit slot, a piece holds total / pieces
    pairs up to one unit (64 pairs), the slab numbering follows the (eighth, offset) groups a piece touches, and every group's slabs are one contiguous
    run that covers exactly the pieces holding its units."""
    import numpy as np
    from seevcn_amd import _lib
    lib = hip_lib
    nbr = _synthetic_lidar_table(n, K, cuda)
    plan = torch.zeros(lib.sv_wgrad_plan_bytes(n, K, pieces) // 4, dtype=torch.int32, device=cuda)
    _lib.check(lib.sv_wgrad_plan_build(nbr.data_ptr(), n, K, pieces, plan.data_ptr(), _lib.stream()), "sv_wgrad_plan_build")
    p = plan.cpu().numpy()
    nbu = -(-(-(-n // 64)) // 8)             # unit slots per group
    G = 8 * K
    U = nbu * G
    pad4 = lambda v: (v + 3) & ~3          # every part of the plan starts on a multiple of 4 ints
    o1 = pad4(pieces + 1)
    o2 = o1 + pad4(pieces + 1)
    o3 = o2 + pad4(2 * G)
    cut, slab0, runs, pre = p[:pieces + 1], p[o1:o1 + pieces + 1], p[o2:o2 + 2 * G].reshape(G, 2), p[o3:o3 + U + 1]
    valid = (nbr >= 0).cpu().numpy()
    per_unit = np.zeros((K, 8 * nbu), dtype=np.int64)                       # pairs of every row unit (slots past the table's end: 0)
    for k in range(K):
        c = np.add.reduceat(valid[k], np.arange(0, n, 64))
        per_unit[k, :len(c)] = c
    counts = per_unit.reshape(K, 8, nbu).transpose(1, 0, 2).reshape(-1)      # (eighth, offset, unit)
    assert np.array_equal(pre, np.concatenate([[0], np.cumsum(counts)]))
    assert cut[0] == 0 and cut[-1] == U and np.all(np.diff(cut) >= 0)
    total = int(counts.sum())
    per_piece = np.array([counts[cut[i]:cut[i + 1]].sum() for i in range(pieces)])
    assert per_piece.sum() == total
    assert np.all(np.abs(per_piece - total / pieces) <= 64 + 1), (per_piece.min(), per_piece.max(), total / pieces)
    nseg = np.array([0 if cut[i] == cut[i + 1] else (cut[i + 1] - 1) // nbu - cut[i] // nbu + 1 for i in range(pieces)])
    assert np.array_equal(slab0, np.concatenate([[0], np.cumsum(nseg)]))
    assert slab0[-1] <= pieces + G - 1
    for g in range(G):
        ids = [slab0[i] + (g - cut[i] // nbu) for i in range(pieces) if cut[i] < cut[i + 1] and cut[i] // nbu <= g <= (cut[i + 1] - 1) // nbu]
        assert ids == list(range(runs[g, 0], runs[g, 0] + runs[g, 1])), g


@pytest.mark.gpu
@pytest.mark.parametrize("n,K,cin,cout", [(100_000, 27, 64, 64), (220_000, 27, 64, 64), (60_000, 27, 64, 128), (90_000, 27, 32, 64), (50_000, 27, 32, 32), (50_000, 27, 16, 32),
                                            (50_000, 27, 16, 16), (40_000, 27, 4, 16), (20_000, 3, 64, 128), (500, 27, 64, 64), (37, 27, 16, 16)])
def test_hip_weight_gradient_on_equal_pieces(cuda, hip_lib, n, K, cin, cout, monkeypatch):
    """sv_sparse_conv_wgrad_planned (every register-tile instance, tables with far fewer units than pieces included) against X[nbr[k]]^T dY in float64, in both
    output layouts, run twice (bitwise the same), and in the two-stage form of the launch list (stage 1 + sv_sparse_conv_wgrad_reduce_batch: bitwise the same).
    sv_wgrad_planned_applies sends only the large 64-channel-multiple layers here by default (where it is faster); SEEVCN_WGRAD_PLANNED_ALL lifts that.
    220 k rows: 2 M pairs = 2 000 per piece, more than a workgroup's list holds (1 536): segments in several chunks."""
    import ctypes
    monkeypatch.setenv("SEEVCN_WGRAD_PLANNED_ALL", "1")
    from seevcn_amd import _lib
    from seevcn_amd.spconv import functional as Fsp
    lib = hip_lib
    nbr = _synthetic_lidar_table(n, K, cuda)
    g = torch.Generator(device=cuda).manual_seed(5)
    x = torch.randn(n, cin, device=cuda, generator=g)
    dy = torch.randn(n, cout, device=cuda, generator=g)
    want = torch.empty((K, cin, cout), dtype=torch.float64, device=cuda)
    for k in range(K):
        rows = torch.nonzero(nbr[k] >= 0).squeeze(1)
        want[k] = x[nbr[k, rows].long()].double().T @ dy[rows].double()
    assert lib.sv_wgrad_planned_applies(n, n, K, cin, cout) == 1
    pieces = lib.sv_wgrad_plan_pieces(cin, cout)
    plan = torch.empty(lib.sv_wgrad_plan_bytes(n, K, pieces), dtype=torch.uint8, device=cuda)
    _lib.check(lib.sv_wgrad_plan_build(nbr.data_ptr(), n, K, pieces, plan.data_ptr(), _lib.stream()), "sv_wgrad_plan_build")
    dw = Fsp.wgrad(x, nbr, dy, K, cin, cout, plan=plan)
    scale = float(want.abs().max())
    assert float((dw.double() - want).abs().max()) <= 2e-5 * scale
    assert torch.equal(Fsp.wgrad(x, nbr, dy, K, cin, cout, plan=plan), dw)
    # the chunked kernel computes the same sums in another order
    assert float((Fsp.wgrad(x, nbr, dy, K, cin, cout) - dw).abs().max()) <= 2e-5 * scale
    like = torch.empty((cout, K, cin), device=cuda).permute(1, 2, 0)          # the parameter's (C_out, k, C_in) memory order
    dws = Fsp.wgrad(x, nbr, dy, K, cin, cout, like=like, plan=plan)
    assert dws.stride() == like.stride() and torch.equal(dws, dw)
    part = torch.empty(lib.sv_sparse_conv_wgrad_planned_bytes(K, cin, cout), dtype=torch.uint8, device=cuda)
    dw2 = torch.empty((K, cin, cout), device=cuda)
    job = (ctypes.c_int64 * 10)()
    rc = lib.sv_sparse_conv_wgrad_planned_stage1(x.data_ptr(), n, nbr.data_ptr(), dy.data_ptr(), dw2.data_ptr(), n, K, cin, cout, cin * cout, cout, 1, plan.data_ptr(),
                                                 part.data_ptr(), job, _lib.stream())
    _lib.check(rc, "sv_sparse_conv_wgrad_planned_stage1")
    assert job[3] > 0 and job[9] != 0
    _lib.check(lib.sv_sparse_conv_wgrad_reduce_batch(job, 1, _lib.stream()), "sv_sparse_conv_wgrad_reduce_batch")
    assert torch.equal(dw2, dw)


@pytest.mark.gpu
def test_hip_weight_gradient_plans_of_several_tables_in_one_batch(cuda, hip_lib):
    """sv_wgrad_plan_build_batch (two launches for all tables) leaves exactly the plans sv_wgrad_plan_build makes one table at a time."""
    import numpy as np
    from seevcn_amd import _lib
    lib = hip_lib
    cases = [(100_000, 27, 1024), (30_000, 27, 2048), (9_000, 3, 1024)]
    tables = [_synthetic_lidar_table(n, K, cuda, seed=3 + q) for q, (n, K, _) in enumerate(cases)]
    single, batch = [], []
    jobs = np.zeros((len(cases), 8), dtype=np.int64)
    for q, ((n, K, pieces), nbr) in enumerate(zip(cases, tables)):
        nb = lib.sv_wgrad_plan_bytes(n, K, pieces)
        a, b = torch.zeros(nb, dtype=torch.uint8, device=cuda), torch.zeros(nb, dtype=torch.uint8, device=cuda)
        _lib.check(lib.sv_wgrad_plan_build(nbr.data_ptr(), n, K, pieces, a.data_ptr(), _lib.stream()), "sv_wgrad_plan_build")
        jobs[q, :5] = (nbr.data_ptr(), n, K, pieces, b.data_ptr())
        single.append(a), batch.append(b)
    _lib.check(lib.sv_wgrad_plan_build_batch(jobs.ctypes.data, len(cases), _lib.stream()), "sv_wgrad_plan_build_batch")
    for a, b in zip(single, batch):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_hip_backward_list_with_weight_gradients_on_a_second_stream_is_bitwise_the_one_stream_list(cuda, hip_lib):
    """sv_run_ops_two_streams (SEEVCN_WGRAD_STREAM=1): the chain's backward with its weight gradients on a side stream -- the same kernels on the same operands,
    ordered behind the chain's stream when the call returns: every gradient bit-identical to the one-stream list."""
    import copy
    import seevcn_amd.synth as synth
    from seevcn_amd.pcdet.models import backbones_3d
    from seevcn_amd.pcdet.ops import voxel_ops
    from seevcn_amd.spconv import chain
    pts, _ = synth.make_scene_batch(2, seed=2006, n_az=120)
    g = dict(point_cloud_range=[0, -40, -3, 70.4, 40, 1], voxel_size=[0.05, 0.05, 0.1], grid_size=[1408, 1600, 40])
    f, c, _ = voxel_ops.voxelize_dynamic(torch.from_numpy(pts).to(cuda), g["point_cloud_range"], g["voxel_size"], g["grid_size"], 2)
    torch.manual_seed(4)
    net1 = backbones_3d.__all__['VoxelBackBone8x']({}, 3, g['grid_size']).to(cuda).train()
    net2 = copy.deepcopy(net1)
    res = []
    for net, two in ((net1, True), (net2, False)):
        saved, chain.WGRAD_STREAM = chain.WGRAD_STREAM, two
        try:
            for _ in range(2):                       # twice: the second pass reuses the events and the side stream
                net.zero_grad(set_to_none=True)
                bd = net({'batch_size': 2, 'voxel_features': f.clone(), 'voxel_coords': c.clone()})
                out = bd['encoded_spconv_tensor'].features
                assert type(out.grad_fn).__name__ == "SparseChainFunctionBackward"
                out.square().sum().backward()
                torch.cuda.synchronize()
        finally:
            chain.WGRAD_STREAM = saved
        res.append({k: p.grad.clone() for k, p in net.named_parameters()})
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,kind", [(16, 16, "subm"), (16, 32, "strided"), (32, 32, "subm"), (32, 64, "strided"), (64, 64, "subm"), (64, 64, "strided"),
                                           (64, 128, "down"), (128, 64, "subm")])
def test_hip_input_norm_on_load_is_bit_identical_to_the_separate_pass(cuda, hip_lib, cin, cout, kind):
    """sv_conv_next_input_norm: the planned conv (forward table and the data-gradient table) and the weight gradients (equal pieces and chunked) read X
    through y = [relu](x * scale + shift) -- against the same kernels on the tensor sv_batchnorm_apply writes: bit-identical, with ReLU and without,
    scales of both signs and a zero scale (an absent neighbour must contribute 0, not relu(shift)); sv_batchnorm_finalize_forward + sv_batchnorm_apply
    against sv_batchnorm_relu_forward (same statistics kernel, same expression); the transform is consumed by ONE call and refused by the plain entry."""
    import seevcn_amd.synth as synth
    from seevcn_amd import _lib
    from seevcn_amd.pcdet.ops import voxel_ops
    from seevcn_amd.spconv import functional as Fsp
    from seevcn_amd.spconv import norm
    lib = hip_lib
    pts, _ = synth.make_scene_batch(2, seed=2003, n_az=100)
    g = dict(point_cloud_range=[0, -40, -3, 70.4, 40, 1], voxel_size=[0.1, 0.1, 0.2], grid_size=[704, 800, 20])
    _, coords, _ = voxel_ops.voxelize_dynamic(torch.from_numpy(pts).to(cuda), g["point_cloud_range"], g["voxel_size"], g["grid_size"], 2)
    shape = [21, 800, 704]
    if kind == "subm":
        rb = Fsp.build_subm_rulebook(coords, 2, shape, [3, 3, 3])
    elif kind == "strided":
        rb = Fsp.build_sparse_rulebook(coords, 2, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    else:
        rb = Fsp.build_sparse_rulebook(coords, 2, shape, [3, 1, 1], [2, 1, 1], [0, 0, 0])
    K = rb.nbr_out.shape[0]
    gen = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn(rb.n_in, cin, generator=gen).to(cuda)
    w = (torch.randn(K, cin, cout, generator=gen) * 0.1).to(cuda)
    dy = torch.randn(rb.n_out, cout, generator=gen).to(cuda)
    scale = torch.randn(cin, generator=gen)
    scale[0] = 0.0                                                                   # gamma == 0: every row's value is relu(shift), an absent neighbour's still 0
    coef = torch.cat([scale, torch.randn(cin, generator=gen) * 0.5]).to(cuda)
    ff, fb = Fsp.fragment_cache.get(w)
    pf = rb.plan("fwd", cin, cout)
    assert pf is not None
    for relu in (1, 0):
        y = torch.empty_like(x)
        _lib.check(lib.sv_batchnorm_apply(_lib.ptr(x), x.shape[0], cin, _lib.ptr(coef), relu, _lib.ptr(y), _lib.stream()), "sv_batchnorm_apply")
        ref = x * coef[:cin] + coef[cin:]
        assert torch.allclose(y, torch.relu(ref) if relu else ref, rtol=1e-6, atol=1e-6)
        want = Fsp.gather_gemm_planned(y, pf, ff, rb.n_out, K, cin, cout)
        lib.sv_conv_next_input_norm(coef.data_ptr(), relu)
        got = Fsp.gather_gemm_planned(x, pf, ff, rb.n_out, K, cin, cout)
        assert torch.equal(got, want), (relu, "forward")
        again = Fsp.gather_gemm_planned(y, pf, ff, rb.n_out, K, cin, cout)                # the transform was consumed: this call reads y as it is
        assert torch.equal(again, want)
        wp = rb.wgrad_plan(cin, cout)
        for plan in ((wp, None) if wp is not None else (None,)):
            want_w = Fsp.wgrad(y, rb.nbr_out, dy, K, cin, cout, plan=plan)
            lib.sv_conv_next_input_norm(coef.data_ptr(), relu)
            got_w = Fsp.wgrad(x, rb.nbr_out, dy, K, cin, cout, plan=plan)
            assert torch.equal(got_w, want_w), (relu, "weight gradient", plan is not None)
    # the data-gradient direction reads gradients, never activations -- but the kernel is the same: the input-major table with a transform on its operand
    pb = rb.plan("bwd", cout, cin)
    if pb is not None:
        coef_o = torch.cat([torch.randn(cout, generator=gen), torch.randn(cout, generator=gen)]).to(cuda)
        yo = torch.empty_like(dy)
        _lib.check(lib.sv_batchnorm_apply(_lib.ptr(dy), dy.shape[0], cout, _lib.ptr(coef_o), 1, _lib.ptr(yo), _lib.stream()), "sv_batchnorm_apply")
        want = Fsp.gather_gemm_planned(yo, pb, fb, rb.n_in, K, cout, cin)
        lib.sv_conv_next_input_norm(coef_o.data_ptr(), 1)
        assert torch.equal(Fsp.gather_gemm_planned(dy, pb, fb, rb.n_in, K, cout, cin), want)
    # finalize + apply == the one-call forward (statistics from x itself), running statistics included
    bn_a, bn_b = torch.nn.BatchNorm1d(cin, eps=1e-3, momentum=0.01).to(cuda).train(), torch.nn.BatchNorm1d(cin, eps=1e-3, momentum=0.01).to(cuda).train()
    with torch.no_grad():
        bn_a.weight.uniform_(-1.5, 1.5), bn_a.bias.uniform_(-0.5, 0.5)
        bn_b.load_state_dict(bn_a.state_dict())
    want_y, want_mean, want_istd = norm.bn_forward_raw(x, bn_a.weight, bn_a.bias, bn_a.running_mean, bn_a.running_var, 0.01, 1e-3, True, True, bn_a.num_batches_tracked)
    coef2, mean, istd, y2 = torch.empty(2 * cin, device=cuda), torch.empty(cin, device=cuda), torch.empty(cin, device=cuda), torch.empty_like(x)
    _lib.check(lib.sv_batchnorm_finalize_forward(_lib.ptr(x), x.shape[0], cin, _lib.ptr(bn_b.weight), _lib.ptr(bn_b.bias), _lib.ptr(bn_b.running_mean), _lib.ptr(bn_b.running_var),
                                                 0.01, 1e-3, _lib.ptr(norm._scratch(cin, cuda)), 0, _lib.ptr(coef2), _lib.ptr(mean), _lib.ptr(istd),
                                                 _lib.ptr(bn_b.num_batches_tracked), _lib.stream()), "sv_batchnorm_finalize_forward")
    _lib.check(lib.sv_batchnorm_apply(_lib.ptr(x), x.shape[0], cin, _lib.ptr(coef2), 1, _lib.ptr(y2), _lib.stream()), "sv_batchnorm_apply")
    assert torch.equal(y2, want_y) and torch.equal(mean, want_mean) and torch.equal(istd, want_istd)
    assert torch.equal(bn_a.running_mean, bn_b.running_mean) and torch.equal(bn_a.running_var, bn_b.running_var) and int(bn_b.num_batches_tracked) == 1
    # the plain k-major entry cannot apply a transform: it must say so instead of ignoring it
    lib.sv_conv_next_input_norm(coef.data_ptr(), 1)
    with pytest.raises(_lib.SeevcnHipError, match="input transform"):
        Fsp.gather_gemm(x, rb.nbr_out, w.permute(0, 2, 1).contiguous(), rb.n_out)
