"""Three graphs of the VCN training step on the same weights and inputs -- (1) the library's own kernels, (2) torch fp32 modules on the GPU (MIOpen /
hipBLASLt), (3) the float64 oracle (oracle/vcn_train.py, pinned to the reference at float64) -- and, per parameter gradient, each fp32 side's error
against float64 in units of the tensor's largest entry, with and without the oracle following that side's decisions inside the 1e-4 band.
Answers "which side owns the difference" (VERDICT round 4, item 1a).  usage: python tools/vcn_train_diag.py [n_objects ...]"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import seevcn_amd  # noqa: E402,F401
import seevcn_amd.synth as synth  # noqa: E402
import seevcn_amd.vcn as V  # noqa: E402
import seevcn_amd.vcn.models.VCN_VC as vc_mod  # noqa: E402
import seevcn_amd.vcn.models.layers as L  # noqa: E402
from seevcn_amd.seeding import seeded_state_dict  # noqa: E402
from oracle import vcn_train as T  # noqa: E402
from test_vcn_train import _device_hints  # noqa: E402

cuda = torch.device("cuda:0")
for n_obj in [int(a) for a in sys.argv[1:]] or [8, 64]:
    for name, seed in (("VCN_VC", 0), ("VCN_CN", 1)):
        clouds, boxes = synth.make_object_batch(n_obj, seed=1000)
        x, bx = torch.from_numpy(clouds).to(cuda), torch.from_numpy(boxes).to(cuda)
        m0 = V.MODELS.build({"NAME": name})
        sd = seeded_state_dict(m0, seed=seed)
        m0.load_state_dict(sd)
        sides = {}
        for side, on_torch in (("own", False), ("torch32", True)):
            m = copy.deepcopy(m0).to(cuda).train()
            saved, vc_mod.TRAIN_ON_TORCH = vc_mod.TRAIN_ON_TORCH, on_torch
            L.TAPS = {}
            try:
                out = m({"input": x, "gt_boxes": bx})
                taps = L.TAPS
            finally:
                vc_mod.TRAIN_ON_TORCH, L.TAPS = saved, None
            up = torch.randn(out["coarse"].shape, generator=torch.Generator().manual_seed(1))
            T.parity_loss(out, up.to(cuda)).backward()
            sides[side] = ({k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None}, taps)

        def oracle(hints):
            fn = T.vcn_vc_train if name == "VCN_VC" else (lambda sd_, c, **kw: T.vcn_cn_train(sd_, c, boxes, **kw))
            outs, leaves, _, over = fn(sd, clouds, hints=hints, band=1e-4 if hints else 0.0)
            T.parity_loss(outs, up).backward()
            return {k: v.grad for k, v in leaves.items()}, over

        free, _ = oracle(None)
        followed, over = oracle(_device_hints(sides["own"][1], clouds.shape[1]))
        print(f"== {name} x {n_obj} objects; oracle followed the own-kernel side at {sum(over.values())} decisions {dict((k, v) for k, v in over.items() if v)}")
        print(f"{'gradient':34s} {'own-f64':>10s} {'own-f64(follow)':>16s} {'torch32-f64':>12s} {'own-torch32':>12s}")
        for k in free:
            s = float(free[k].abs().max()) or 1.0
            e = [float((sides["own"][0][k] - free[k]).abs().max()) / s, float((sides["own"][0][k] - followed[k]).abs().max()) / s,
                 float((sides["torch32"][0][k] - free[k]).abs().max()) / s, float((sides["own"][0][k] - sides["torch32"][0][k]).abs().max()) / s]
            print(f"{k:34s} {e[0]:10.2e} {e[1]:16.2e} {e[2]:12.2e} {e[3]:12.2e}" + ("   <-- > 1e-3" if max(e) > 1e-3 else ""))
