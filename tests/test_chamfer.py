"""VCN training loss pieces: Chamfer distance forward/backward, batch FPS + gather, VCN_VC train-mode forward
and get_loss.  The Chamfer / FPS kernels of the reference are CUDA-only (parity unpinned, see oracle/chamfer.py); the Python
layers above them are pinned by tests/golden/vcn_loss.npz = the reference's own VCN_VC.forward (train mode) and get_loss run on
CPU with those two ops served by the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import chamfer as och
from oracle import pointnet2 as op2
from seeding import seeded_state_dict
from tolerances import assert_close_per_channel
from vcn_loss_inputs import make_batch


def test_oracle_chamfer_vs_ckdtree():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=(3, 700, 3)).astype(np.float32), rng.normal(size=(3, 1300, 3)).astype(np.float32)
    d1, d2, i1, i2 = och.forward(a, b)
    for k in range(3):
        dd, ii = cKDTree(b[k].astype(np.float64)).query(a[k].astype(np.float64))
        assert np.array_equal(ii, i1[k]) and np.allclose(dd ** 2, d1[k], rtol=1e-5, atol=1e-7)
        dd, ii = cKDTree(a[k].astype(np.float64)).query(b[k].astype(np.float64))
        assert np.array_equal(ii, i2[k]) and np.allclose(dd ** 2, d2[k], rtol=1e-5, atol=1e-7)


def test_vcn_train_forward_and_loss_match_reference_golden_cpu(golden_dir, monkeypatch):
    """The torch-module statement of the train-mode forward (SEEVCN_VCN_TRAIN_TORCH: the cross-check of the own-kernel path, tests/test_dense_ops.py)
    against the reference's own train-mode forward on CPU; the three losses that need no CUDA op (dims, translation, rotation) against the
    reference's get_loss.  The default train-mode forward runs on the library's kernels and refuses CPU tensors."""
    import seevcn_amd.vcn as V
    import seevcn_amd.vcn.models.VCN_VC as vc_mod
    from seevcn_amd import _lib
    g = np.load(os.path.join(golden_dir, "vcn_loss.npz"))
    inp, complete, gt = make_batch()
    m = V.MODELS.build({"NAME": "VCN_VC"})
    m.load_state_dict(seeded_state_dict(m, seed=0))
    m.train()
    with pytest.raises(_lib.SeevcnHipError):
        m({"input": torch.from_numpy(inp)})
    monkeypatch.setattr(vc_mod, "TRAIN_ON_TORCH", True)
    ret = m({"input": torch.from_numpy(inp)})
    for k in ("coarse", "reg_rot", "reg_centre"):
        assert np.abs(ret[k].detach().numpy() - g[k]).max() <= 1e-3 * np.abs(g[k]).max() + 1e-5, k
    ld = m.get_loss(ret, {"gt_boxes": torch.from_numpy(gt), "training": False})
    for k in ("dims", "translation", "rotation"):
        assert abs(float(ld[k]) - float(g["loss_" + k])) <= 1e-3 * abs(float(g["loss_" + k])) + 1e-6, k


def test_vcn_cn_train_forward_matches_reference_golden_cpu(golden_dir, monkeypatch):
    import seevcn_amd.vcn as V
    import seevcn_amd.vcn.models.VCN_VC as vc_mod
    g = np.load(os.path.join(golden_dir, "vcn_loss.npz"))
    inp, complete, gt = make_batch()
    m = V.MODELS.build({"NAME": "VCN_CN"})
    m.load_state_dict(seeded_state_dict(m, seed=1))
    m.train()
    monkeypatch.setattr(vc_mod, "TRAIN_ON_TORCH", True)       # the torch-module statement; the default (own kernels) is GPU-only
    ret = m({"input": torch.from_numpy(inp), "gt_boxes": torch.from_numpy(gt)})
    assert np.abs(ret["coarse"].detach().numpy() - g["cn_coarse"]).max() <= 1e-3 * np.abs(g["cn_coarse"]).max() + 1e-5
    assert m.get_loss(ret, {"gt_boxes": torch.from_numpy(gt), "training": False}) == {}


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_vcn_cn_get_loss_matches_reference_golden(golden_dir, cuda, hip_lib):
    import seevcn_amd.vcn as V
    g = np.load(os.path.join(golden_dir, "vcn_loss.npz"))
    inp, complete, gt = make_batch()
    m = V.MODELS.build({"NAME": "VCN_CN"})
    m.load_state_dict(seeded_state_dict(m, seed=1))
    m = m.to(cuda).train()
    dev = lambda a: torch.from_numpy(a).to(cuda)
    ret = m({"input": dev(inp), "gt_boxes": dev(gt)})
    assert_close_per_channel(ret["coarse"].detach().cpu().numpy(), g["cn_coarse"], rtol=1e-3, atol_frac=1e-4, name="cn_coarse")
    ld = m.get_loss(ret, {"gt_boxes": dev(gt), "training": True, "complete": dev(complete), "input": dev(inp)})
    assert abs(float(ld["coarse"]) - float(g["cn_loss_coarse"])) <= 1e-3 * abs(float(g["cn_loss_coarse"]))
    assert torch.isfinite(ld["partial"]) and float(ld["partial"]) >= 0
    ld["coarse"].backward()
    assert float(m.shape_fc[0].weight.grad.abs().sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,n,m", [(3, 700, 1300), (2, 1024, 1024), (1, 1, 5), (4, 513, 512)])
def test_hip_chamfer_forward_backward_vs_oracle(cuda, hip_lib, B, n, m):
    from seevcn_amd.vcn.extensions.chamfer_dist import ChamferDistanceL1, ChamferDistanceL2, ChamferFunction
    rng = np.random.default_rng(B * 1000 + n)
    a, b = rng.normal(size=(B, n, 3)).astype(np.float32), rng.normal(size=(B, m, 3)).astype(np.float32)
    b[0, m // 2] = b[0, 0]                                                     # exact tie: the first index must win
    ta, tb = torch.from_numpy(a).to(cuda).requires_grad_(True), torch.from_numpy(b).to(cuda).requires_grad_(True)
    d1, d2 = ChamferFunction.apply(ta, tb)
    o1, o2, i1, i2 = och.forward(a, b)
    assert np.array_equal(d1.detach().cpu().numpy(), o1) and np.array_equal(d2.detach().cpu().numpy(), o2)    # fp32, same order: bit-exact
    w1, w2 = rng.normal(size=(B, n)).astype(np.float32), rng.normal(size=(B, m)).astype(np.float32)
    (d1 * torch.from_numpy(w1).to(cuda)).sum().add((d2 * torch.from_numpy(w2).to(cuda)).sum()).backward()
    g1, g2 = och.backward(a, b, i1, i2, w1, w2)
    np.testing.assert_allclose(ta.grad.cpu().numpy(), g1, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), g2, rtol=1e-4, atol=1e-4)
    l2 = float(ChamferDistanceL2()(ta, tb))
    assert abs(l2 - (o1.mean() + o2.mean())) <= 1e-5 * abs(l2)
    l1 = float(ChamferDistanceL1()(ta, tb))
    assert abs(l1 - (np.sqrt(o1).mean() + np.sqrt(o2).mean()) / 2) <= 1e-5 * abs(l1)


@pytest.mark.gpu
def test_hip_vcn_get_loss_matches_reference_golden(golden_dir, cuda, hip_lib):
    import seevcn_amd.vcn as V
    from seevcn_amd.vcn.utils import misc
    g = np.load(os.path.join(golden_dir, "vcn_loss.npz"))
    inp, complete, gt = make_batch()
    m = V.MODELS.build({"NAME": "VCN_VC"})
    m.load_state_dict(seeded_state_dict(m, seed=0))
    m = m.to(cuda).train()
    ret = m({"input": torch.from_numpy(inp).to(cuda)})
    for k in ("coarse", "reg_rot", "reg_centre"):
        assert_close_per_channel(ret[k].detach().cpu().numpy(), g[k], rtol=1e-3, atol_frac=1e-4, name="vcn_train_" + k)
    ds = misc.fps(torch.from_numpy(complete).to(cuda), 1024)
    idx = np.stack([op2.farthest_point_sampling(complete[b], 1024) for b in range(len(complete))])
    assert np.array_equal(ds.cpu().numpy(), np.stack([complete[b][idx[b]] for b in range(len(complete))]))
    ld = m.get_loss(ret, {"gt_boxes": torch.from_numpy(gt).to(cuda), "training": True, "complete": torch.from_numpy(complete).to(cuda),
                          "input": torch.from_numpy(inp).to(cuda)})
    for k in ("dims", "translation", "rotation", "coarse"):
        assert abs(float(ld[k]) - float(g["loss_" + k])) <= 1e-3 * abs(float(g["loss_" + k])) + 1e-6, (k, float(ld[k]), float(g["loss_" + k]))
    assert torch.isfinite(ld["partial"]) and float(ld["partial"]) >= 0
    total = sum(ld[k] for k in ("dims", "translation", "rotation", "coarse"))
    total.backward()
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in m.parameters())
    assert m.shape_fc[0].weight.grad is not None and float(m.shape_fc[0].weight.grad.abs().sum()) > 0
