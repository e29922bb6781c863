"""VCN_VC / VCN_CN: oracle vs the reference's goldens (CPU); HIP path vs goldens and oracle (GPU).
Tolerance: 1e-3 relative on feature/point tensors (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from oracle import vcn as ovcn
from seeding import seeded_state_dict

RTOL = 1e-3


def _ok(a, b, rtol=RTOL, atol_frac=1e-4, name=""):
    """element-wise |a-b| <= rtol*|b| + atol_frac * max|b[..., c]| per coordinate / channel (tests/tolerances.py)"""
    from tolerances import assert_close_per_channel
    assert_close_per_channel(a, b, rtol=rtol, atol_frac=atol_frac, name=name)
    return True


def _rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def _models():
    import seevcn_amd.vcn as V
    return V


def test_registry_and_state_dict_keys():
    V = _models()
    assert set(V.MODELS.module_dict) >= {"VCN_VC", "VCN_CN"}
    m = V.MODELS.build({"NAME": "VCN_VC"})
    assert sum(p.numel() for p in m.parameters()) == 7534476
    keys = list(m.state_dict())
    assert "pose_encoder.4.weight" in keys and "encoder.mlp_conv2.1.running_var" in keys and "final_conv.6.bias" in keys
    with pytest.raises(KeyError):
        V.MODELS.build({"NAME": "nope"})
    with pytest.raises(KeyError):
        V.MODELS.build({})


def test_oracle_vcn_vc_matches_reference_golden(golden_dir):
    V = _models()
    g = np.load(os.path.join(golden_dir, "vcn_vc.npz"))
    sd = seeded_state_dict(V.MODELS.build({"NAME": "VCN_VC"}), seed=0)
    out = ovcn.vcn_vc_forward(sd, torch.from_numpy(g["input"]))
    for k in ("coarse", "reg_rot", "reg_centre"):
        assert _rel_err(out[k].numpy(), g[k]) < 1e-4, k


def test_oracle_vcn_cn_matches_reference_golden(golden_dir):
    V = _models()
    g = np.load(os.path.join(golden_dir, "vcn_cn.npz"))
    sd = seeded_state_dict(V.MODELS.build({"NAME": "VCN_CN"}), seed=0)
    out = ovcn.vcn_cn_forward(sd, torch.from_numpy(g["input"]), torch.from_numpy(g["gt_boxes"]))
    assert _rel_err(out["coarse"].numpy(), g["coarse"]) < 1e-4


def test_eval_forward_refuses_cpu_tensors():
    V = _models()
    import seevcn_amd._lib as L
    m = V.MODELS.build({"NAME": "VCN_VC"})
    with pytest.raises(L.SeevcnHipError):
        m.eval()({"input": torch.zeros(1, 1024, 3)})  # the HIP inference path has no CPU fallback
    with pytest.raises(L.SeevcnHipError):                # nor have the loss ops of the (torch-autograd) training path
        m.get_loss({"coarse": torch.zeros(1, 1024, 3), "reg_rot": torch.eye(3)[None], "reg_centre": torch.zeros(1, 3)},
                   {"gt_boxes": torch.ones(1, 7), "training": True, "complete": torch.zeros(1, 2048, 3), "input": torch.zeros(1, 1024, 3)})


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_gemm_epilogues(cuda, hip_lib):
    """sv_gemm_bias_act against a torch fp32 reference: masks on M/N, bias, group bias, activations, group max."""
    from seevcn_amd.vcn.models import layers as L
    g = torch.Generator().manual_seed(0)
    for (M, N, K, rpg) in [(256, 128, 64, 128), (300, 9, 32, 100), (1024, 200, 96, 1024), (64, 3072, 1024, 1), (2048, 256, 128, 512)]:
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        ng = (M + rpg - 1) // rpg
        gb = torch.randn(ng, N, generator=g)
        ref = a.double() @ w.double().t() + b.double() + gb.double().repeat_interleave(rpg, 0)[:M]
        for act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU):
            r = ref.clone()
            if act == L.ACT_RELU:
                r = r.clamp_min(0)
            elif act == L.ACT_LRELU:
                r = torch.where(r >= 0, r, r * 0.01)
            gm = L.neg_inf((ng, N), cuda)
            out = L.gemm(a.to(cuda), w.to(cuda), b.to(cuda), act, group_bias=gb.to(cuda), rows_per_group=rpg, group_max=gm)
            torch.cuda.synchronize()
            np.testing.assert_allclose(out.cpu().numpy(), r.numpy(), rtol=1e-4, atol=1e-4)
            rmax = torch.stack([r[i * rpg:(i + 1) * rpg].max(0)[0] for i in range(ng)])
            np.testing.assert_allclose(gm.cpu().numpy(), rmax.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_hip_gemm_small_m_kernel(cuda, hip_lib):
    """M <= 64 without group options runs k_gemm_small_m (16 columns per workgroup, K split over the waves): every M / N / K edge,
    bias or not, all activations, against float64; and the same rows through the 128x128-tile kernel agree to fp32 rounding."""
    from seevcn_amd.vcn.models import layers as L
    g = torch.Generator().manual_seed(1)
    for (M, N, K) in [(64, 1024, 1024), (1, 9, 512), (5, 3072, 1024), (64, 9, 32), (33, 40, 64), (17, 512, 256)]:
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        for bias in (b, None):
            ref = a.double() @ w.double().t() + (bias.double() if bias is not None else 0)
            for act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU):
                r = ref.clamp_min(0) if act == L.ACT_RELU else (torch.where(ref >= 0, ref, ref * 0.01) if act == L.ACT_LRELU else ref)
                out = L.gemm(a.to(cuda), w.to(cuda), bias.to(cuda) if bias is not None else None, act)
                np.testing.assert_allclose(out.cpu().numpy(), r.numpy(), rtol=1e-4, atol=1e-4)
        # 65 rows take the tile kernel: its first 64 rows must match the small-M result closely (different summation order)
        a65 = torch.cat([a, a[:1]], 0) if M == 64 else None
        if a65 is not None:
            big = L.gemm(a65.to(cuda), w.to(cuda), b.to(cuda), L.ACT_NONE)[:64]
            small = L.gemm(a.to(cuda), w.to(cuda), b.to(cuda), L.ACT_NONE)
            np.testing.assert_allclose(big.cpu().numpy(), small.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_hip_vcn_vc_matches_reference_golden(golden_dir, cuda, hip_lib):
    V = _models()
    g = np.load(os.path.join(golden_dir, "vcn_vc.npz"))
    m = V.MODELS.build({"NAME": "VCN_VC"})
    m.load_state_dict(seeded_state_dict(m, seed=0))
    m = m.to(cuda).eval()
    out = m({"input": torch.from_numpy(g["input"]).to(cuda)})
    torch.cuda.synchronize()
    for k in ("coarse", "reg_rot", "reg_centre"):
        assert out[k].shape == g[k].shape
        assert _ok(out[k].cpu().numpy(), g[k], name=k)


@pytest.mark.gpu
def test_hip_vcn_cn_matches_reference_golden(golden_dir, cuda, hip_lib):
    V = _models()
    g = np.load(os.path.join(golden_dir, "vcn_cn.npz"))
    m = V.MODELS.build({"NAME": "VCN_CN"})
    m.load_state_dict(seeded_state_dict(m, seed=0))
    m = m.to(cuda).eval()
    out = m({"input": torch.from_numpy(g["input"]).to(cuda), "gt_boxes": torch.from_numpy(g["gt_boxes"]).to(cuda)})
    torch.cuda.synchronize()
    assert _ok(out["coarse"].cpu().numpy(), g["coarse"], name="coarse")


@pytest.mark.gpu
def test_hip_vcn_vc_batch64_vs_oracle_and_batch_invariance(cuda, hip_lib):
    """BASELINE config 2 size (64 objects x 1024 pts): vs oracle, and each object's result must not depend on its batch."""
    import seevcn_amd.synth as synth
    V = _models()
    clouds, _ = synth.make_object_batch(64, seed=1000)
    m = V.MODELS.build({"NAME": "VCN_VC"})
    sd = seeded_state_dict(m, seed=0)
    m.load_state_dict(sd)
    m = m.to(cuda).eval()
    x = torch.from_numpy(clouds).to(cuda)
    out = m({"input": x})
    ref = ovcn.vcn_vc_forward(sd, torch.from_numpy(clouds[:8]))
    for k in ("coarse", "reg_rot", "reg_centre"):
        assert _ok(out[k][:8].cpu().numpy(), ref[k].numpy(), name=k)
    sub = m({"input": x[5:8].contiguous()})
    assert _rel_err(sub["coarse"].cpu().numpy(), out["coarse"][5:8].cpu().numpy()) < 1e-5
    # padded zero objects (VCN.inference pads chunks with zeros, models/VCN.py:55-59) must not produce NaN
    z = m({"input": torch.zeros(2, 1024, 3, device=cuda)})
    assert torch.isfinite(z["coarse"]).all()


def test_resample_points_contract():
    from seevcn_amd.vcn.datasets.data_transforms import ResamplePoints
    pts = np.arange(90, dtype=np.float64).reshape(30, 3)
    np.random.seed(0)
    out = ResamplePoints({"n_points": 1024})(pts)
    np.random.seed(0)
    ref = np.tile(pts, (35, 1))[np.random.permutation(1050)[:1024]]          # data_transforms.py:254-262
    assert out.shape == (1024, 3) and np.array_equal(out, ref)
    big = np.random.default_rng(0).normal(size=(5000, 3))
    assert ResamplePoints({"n_points": 1024})(big).shape == (1024, 3)


@pytest.mark.gpu
def test_hip_vcn_inference_wrapper_chunking(cuda, hip_lib):
    """VCN.inference: resample -> pad to BATCH_SIZE_LIMIT -> chunks -> first num_objs rows (models/VCN.py:43-83)."""
    import seevcn_amd.synth as synth
    from seevcn_amd.vcn.VCN import VCN
    V = _models()
    sd = seeded_state_dict(V.MODELS.build({"NAME": "VCN_VC"}), seed=0)
    vcn = VCN({"MODEL": "VCN_VC", "NORM_WITH_GT": False, "SEL_K_NEAREST": 20, "CLUSTER_EPS": 0.3, "BATCH_SIZE_LIMIT": 4}, 0,
              state_dict={"module." + k: v for k, v in sd.items()})
    objs = [synth.make_object(np.random.default_rng(1000 + i))[0] for i in range(6)]
    np.random.seed(3)
    out = vcn.inference(objs, batch_size_limit=4)
    assert out["input"].shape == (6, 1024, 3) and out["coarse"].shape == (6, 1024, 3)
    ref = ovcn.vcn_vc_forward(sd, torch.from_numpy(out["input"]))["coarse"].numpy()
    assert _ok(out["coarse"], ref, name="coarse")
    # post-processing on the GPU's own coarse output (models/VCN.py:89-93) against the CPU restatement
    from oracle import postprocess as opp
    surf = opp.get_partial_mesh_batch(out["input"], out["coarse"], k=30)
    assert out["surface"].dtype == np.float32 and np.array_equal(out["surface"], surf)
    assert out["clustered"].dtype == np.float64 and np.array_equal(out["clustered"], opp.get_largest_cluster_batch(surf, eps=0.4, min_points=2))
    np.random.seed(3)
    single = vcn.inference(objs[0])
    assert np.array_equal(single["input"][0], out["input"][0]) and _ok(single["coarse"][0], ref[0], name="single coarse")


@pytest.mark.gpu
def test_hip_vcn_distinct_row_path_is_bit_identical(cuda, hip_lib):
    """ResamplePoints tiles Ni points to 1024: the per-point layers run on the distinct rows only (sv_unique_rows +
    sv_gemm_bias_act_ragged).  Output must equal the full 1024-row execution bit for bit, for VCN_VC and VCN_CN, including an
    all-zero padding object (one distinct row) and an object with 1024 distinct points."""
    import seevcn_amd.synth as synth
    V = _models()
    clouds, boxes = synth.make_object_batch(6, seed=1000)
    clouds[4] = 0.0
    clouds[5] = np.random.default_rng(0).normal(size=(1024, 3)).astype(np.float32) + np.array([20, 3, -1], np.float32)
    x, bx = torch.from_numpy(clouds).to(cuda), torch.from_numpy(boxes).to(cuda)
    for name, extra in (("VCN_VC", {}), ("VCN_CN", {"gt_boxes": bx})):
        m = V.MODELS.build({"NAME": name})
        m.load_state_dict(seeded_state_dict(m, seed=0))
        m = m.to(cuda).eval()
        m.dedup_points = True
        a = m({"input": x, **extra})                      # default: the distinct-row count never leaves the device (capacity-sized launches)
        m.dedup_points = False
        b = m({"input": x, **extra})
        for k in a:
            assert torch.equal(a[k], b[k]), (name, k)
        if name == "VCN_VC":
            import seevcn_amd.vcn.models.VCN_VC as vc_mod
            m.dedup_points = True
            saved, vc_mod.LAZY_ROWS = vc_mod.LAZY_ROWS, False
            try:
                c = m({"input": x, **extra})                  # the count read on the host, exact-size layers
            finally:
                vc_mod.LAZY_ROWS = saved
            for k in a:
                assert torch.equal(a[k], c[k]), (name, k, "host-read row count")
    from seevcn_amd.vcn.models import layers as L
    sel, rg = L.distinct_rows(x)
    sel_cap, rg_cap, u_dev = L.distinct_rows(x, sync=False)
    assert int(u_dev) == sel.shape[0] and sel_cap.shape[0] == x.shape[0] * x.shape[1]
    assert torch.equal(sel_cap[:sel.shape[0]], sel) and torch.equal(rg_cap[:sel.shape[0]], rg)
    counts = torch.bincount(rg.long(), minlength=6).cpu().numpy()
    want = [len(np.unique(clouds[i], axis=0)) for i in range(6)]
    assert counts.tolist() == want and counts[4] == 1 and counts[5] == 1024
