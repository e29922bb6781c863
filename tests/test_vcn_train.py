"""VCN_VC / VCN_CN in TRAINING mode (batch-statistics BatchNorm), forward + hand-written backward on the library's own kernels (dense_ops.linear,
fused BatchNorm + ReLU, segment_max -- see-vcn_amd/vcn/models/layers.py encode_train) against the float64 oracle oracle/vcn_train.py, which is itself
pinned to the REFERENCE's modules run at float64 (tests/golden/vcn_train.npz, make_vcn_train_golden.py).  Reference: VCN_VC.py:97-106,178-214,
VCN_CN.py:142-156.

Tolerances (written here): outputs and running statistics 1e-3 relative per element + 1e-4 of the tensor's largest entry (north_star: "within 1e-3
rel on ... feature tensors"); every parameter gradient within 1e-3 of that tensor's largest entry (a gradient entry is a sum of up to 65 536 fp32
products: its error scales with the tensor, not with the entry).  The oracle follows the device's ReLU / LeakyReLU / arg-max decision only inside a
band of 1e-4 x RMS around the decision boundary; the number of followed decisions is capped at 2e-5 of all decisions (rounding noise near zero), so a
kernel that takes wrong branches fails the cap instead of being excused."""
import numpy as np
import pytest
import torch

BAND = 1e-4
ZERO_GRADS = ("encoder.mlp_conv1.0.bias", "encoder.mlp_conv1.3.bias", "encoder.mlp_conv2.0.bias")   # a constant per channel in front of a batch-statistics BatchNorm:
#                                                                                                       exactly zero gradient, rounding noise on every side


def _build(name, seed):
    import seevcn_amd.vcn as V
    from seeding import seeded_state_dict
    m = V.MODELS.build({"NAME": name})
    sd = seeded_state_dict(m, seed=seed)
    m.load_state_dict(sd)
    return m, sd


def _oracle(name, sd, clouds, boxes, hints=None, band=0.0):
    from oracle import vcn_train as T
    if name == "VCN_VC":
        outs, leaves, bufs, over = T.vcn_vc_train(sd, clouds, hints=hints, band=band)
    else:
        outs, leaves, bufs, over = T.vcn_cn_train(sd, clouds, boxes, hints=hints, band=band)
    up = torch.randn(outs["coarse"].shape, generator=torch.Generator().manual_seed(1))
    T.parity_loss(outs, up).backward()
    return outs, leaves, bufs, over, up


@pytest.mark.parametrize("name,tag,seed", [("VCN_VC", "vc", 0), ("VCN_CN", "cn", 1)])
def test_oracle_training_graph_matches_reference_float64_golden(golden_dir, name, tag, seed):
    """oracle/vcn_train.py (float64, no hints) == the reference's own modules at float64: outputs, sampled gradients of every parameter, running stats"""
    from oracle import vcn_train as T
    g = np.load(f"{golden_dir}/vcn_train.npz")
    _, sd = _build(name, seed)
    outs, leaves, bufs, over, _ = _oracle(name, sd, g["input"], g["gt_boxes"])
    assert not over
    for k, v in outs.items():
        np.testing.assert_allclose(v.detach().numpy(), g[f"{tag}.out.{k}"], rtol=1e-10, atol=1e-12)
    keys = sorted(k[len(tag) + 6:] for k in g.files if k.startswith(tag + ".grad."))
    assert keys == sorted(leaves) and len(keys) >= 10
    gmax = max(float(g[f"{tag}.gmax.{k}"]) for k in keys)
    for k in keys:
        gr = leaves[k].grad.reshape(-1)
        got, want = gr[T.sample_index(gr.numel())].numpy(), g[f"{tag}.grad.{k}"]
        if k in ZERO_GRADS:
            assert np.abs(got).max() <= 1e-9 * gmax and np.abs(want).max() <= 1e-9 * gmax
            continue
        assert np.abs(got - want).max() <= 1e-10 * float(g[f"{tag}.gmax.{k}"]), k
    n_buf = 0
    for k in g.files:
        if k.startswith(tag + ".buf.") and k[len(tag) + 5:] in bufs:
            np.testing.assert_allclose(bufs[k[len(tag) + 5:]].numpy(), g[k], rtol=1e-12, atol=1e-14)
            n_buf += 1
    assert n_buf == 6                                                                   # mean / var / count of the encoder's two norms


def _device_hints(taps, n):
    hints = {}
    for k, v in taps.items():
        if isinstance(v, tuple):                                                        # (pool input (B n, C), pool output (B, C)) -> chosen row per (object, channel)
            z, out = v
            zv = z.view(-1, n, z.shape[1])
            hit = zv == out[:, None, :]
            assert bool(hit.any(dim=1).all()), k                                        # the pool's output IS one of its inputs
            hints[k] = hit.int().argmax(dim=1).cpu()
        else:
            hints[k] = (v > 0).cpu()
    return hints


@pytest.mark.gpu
@pytest.mark.parametrize("name,seed,n_obj", [("VCN_VC", 0, 8), ("VCN_CN", 1, 8), ("VCN_VC", 0, 64), ("VCN_CN", 1, 64)])
def test_hip_vcn_training_forward_backward_vs_float64_oracle(cuda, hip_lib, golden_dir, name, seed, n_obj):
    """8 objects = the golden's inputs (also compared with the reference's float64 numbers directly); 64 objects x 1024 points = BASELINE configs[1]"""
    import seevcn_amd.synth as synth
    import seevcn_amd.vcn.models.layers as L
    from oracle import vcn_train as T
    from oracle.tolerances import assert_close_per_channel
    clouds, boxes = synth.make_object_batch(n_obj, seed=1000)
    m, sd = _build(name, seed)
    m = m.to(cuda).train()
    L.TAPS = {}
    try:
        out = m({"input": torch.from_numpy(clouds).to(cuda), "gt_boxes": torch.from_numpy(boxes).to(cuda)})
        taps = L.TAPS
    finally:
        L.TAPS = None
    up = torch.randn(out["coarse"].shape, generator=torch.Generator().manual_seed(1))
    T.parity_loss(out, up.to(cuda)).backward()
    grads = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}
    hints = _device_hints(taps, clouds.shape[1])
    assert sorted(hints) == sorted(["enc.act1", "enc.max1", "enc.act2", "enc.max2", "shape_fc.act0", "shape_fc.act1"]
                                   + (["pose.act0", "pose.act1", "pose.max", "pose_fc.act0"] if name == "VCN_VC" else []))
    outs, leaves, bufs, over, _ = _oracle(name, sd, clouds, boxes, hints=hints, band=BAND)
    decisions = sum(int(h.numel()) for h in hints.values())
    assert sum(over.values()) <= max(8, 2e-5 * decisions), (over, decisions)
    for k in outs:
        assert_close_per_channel(out[k].detach().cpu().numpy(), outs[k].detach().numpy(), rtol=1e-3, atol_frac=1e-4, name=f"{name} {k}")
    assert sorted(grads) == sorted(leaves) and len(grads) >= 10
    gmax = max(float(v.grad.abs().max()) for v in leaves.values())
    worst = {}
    for k, want in leaves.items():
        got, want = grads[k], want.grad
        if k in ZERO_GRADS:
            assert float(got.abs().max()) <= 1e-5 * gmax, (k, float(got.abs().max()), gmax)
            continue
        err, scale = float((got - want).abs().max()), float(want.abs().max())
        worst[k] = err / scale
        assert err <= 1e-3 * scale, (name, k, err, scale, over)
    for k, b in m.named_buffers():
        if k in bufs:
            assert_close_per_channel(b.detach().double().cpu().numpy(), bufs[k].numpy(), rtol=1e-3, atol_frac=1e-4, name=f"{name} buffer {k}")
    if n_obj == 8:                                                                       # the golden's batch: the reference's own float64 gradients, no oracle in between
        g, tag = np.load(f"{golden_dir}/vcn_train.npz"), "vc" if name == "VCN_VC" else "cn"
        assert np.array_equal(g["input"], clouds)
        for k in out:
            assert_close_per_channel(out[k].detach().cpu().numpy(), g[f"{tag}.out.{k}"], rtol=1e-3, atol_frac=1e-4, name=f"{name} {k} vs reference")
        if sum(over.values()) == 0:                                                      # no decision inside rounding distance of its boundary: the graphs are the same
            for k in grads:
                if k in ZERO_GRADS:
                    continue
                gr = grads[k].reshape(-1)
                assert float((gr[T.sample_index(gr.numel())] - torch.from_numpy(g[f"{tag}.grad.{k}"])).abs().max()) <= 1e-3 * float(g[f"{tag}.gmax.{k}"]), k
    print(f"{name} x{n_obj}: followed {sum(over.values())} of {decisions} decisions; worst gradient error / scale {max(worst.values()):.2e} ({max(worst, key=worst.get)})")


@pytest.mark.gpu
def test_hip_vcn_training_with_a_frozen_norm_takes_the_module_path(cuda, hip_lib):
    """A BatchNorm put in eval() inside a training model (fine-tuning with frozen statistics) is not fusable in the differentiable path: it runs as the
    module and the backward still works."""
    import seevcn_amd.synth as synth
    m, _ = _build("VCN_CN", 1)
    m = m.to(cuda).train()
    m.encoder.mlp_conv1[1].eval()
    clouds, boxes = synth.make_object_batch(4, seed=1000)
    out = m({"input": torch.from_numpy(clouds).to(cuda), "gt_boxes": torch.from_numpy(boxes).to(cuda)})
    out["coarse"].sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    assert int(m.encoder.mlp_conv1[1].num_batches_tracked) == 0 and int(m.encoder.mlp_conv2[1].num_batches_tracked) == 1
