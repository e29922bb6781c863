"""Oracle: VCN_VC / VCN_CN in TRAINING mode (batch-statistics BatchNorm) as one differentiable chain of plain tensor algebra on the CPU, float64 by
default.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows see/surface_completion/models/vcn/models/VCN_VC.py:97-106 (FeatureEncoder.forward), :178-214 (VCN_VC.forward), :12-49 (6-D rotation),
models/VCN_CN.py:142-156 with utils/transform.py:33-57,91-160.  BatchNorm1d in train mode = torch's definition: normalise with the batch mean and
the BIASED batch variance, eps 1e-5; running_mean / running_var move by momentum 0.1 towards the batch mean / the UNBIASED batch variance.

PINNED: tests/golden/vcn_train.npz holds the reference's own VCN_VC / VCN_CN modules run in train mode at float64 (tests/golden/
make_vcn_train_golden.py): outputs, sampled parameter gradients, running statistics; tests/test_vcn_train.py checks this file against it on the CPU.

Branch hints.  The graph has discrete decisions -- ReLU / LeakyReLU branches and the arg-max of the pools over an object's points.  For a
pre-activation within rounding distance of zero (or two pool candidates within rounding distance of each other) a float32 and a float64 evaluation
may decide differently, both valid, and the gradients downstream then differ by O(1) in a few entries.  `hints` {decision name: tensor} are the
decisions the evaluation under test took; a hint is followed ONLY inside a band of `band` x the RMS of the tensor the decision is taken on, everywhere
else the oracle's own value decides.  The third return value counts the positions where a hint changed the oracle's decision (tests cap it).
Decision names: 'pose.act0', 'pose.act1', 'pose.max', 'pose_fc.act0', 'enc.act1', 'enc.max1', 'enc.act2', 'enc.max2', 'shape_fc.act0',
'shape_fc.act1' -- act*: bool (rows, C) "output > 0"; *max*: int64 (B, C) index of the chosen point inside its object."""
import torch


class _Hints:
    def __init__(self, hints, band):
        self.hints, self.band, self.overridden = hints or {}, float(band), {}

    def act(self, name, z, slope):
        """z (rows, C) pre-activation -> ReLU (slope 0) / LeakyReLU(slope)"""
        on = z.detach() > 0
        h = self.hints.get(name)
        if h is not None:
            h = torch.as_tensor(h).reshape(on.shape).bool()
            use = (z.detach().abs() < self.band * z.detach().pow(2).mean().sqrt()) & (h != on)
            self.overridden[name] = int(use.sum())
            on = torch.where(use, h, on)
        return torch.where(on, z, z * slope)

    def group_max(self, name, z, n):
        """z (B n, C) -> max over each object's n rows (B, C), gradient to the chosen row"""
        B, C = z.shape[0] // n, z.shape[1]
        zv = z.view(B, n, C)
        top, arg = zv.detach().max(dim=1)
        h = self.hints.get(name)
        if h is not None:
            h = torch.as_tensor(h).reshape(B, C).long()
            at_hint = zv.detach().gather(1, h[:, None, :])[:, 0]
            use = (top - at_hint < self.band * z.detach().pow(2).mean().sqrt()) & (h != arg) & (at_hint != top)   # exact ties: duplicated points, same gradient sums
            self.overridden[name] = int(use.sum())
            arg = torch.where(use, h, arg)
        return zv.gather(1, arg[:, None, :])[:, 0]


def _rotz(points, angle):
    """rotate_points_along_z, utils/transform.py:33-57: p @ [[c,s,0],[-s,c,0],[0,0,1]]"""
    c, s = torch.cos(angle), torch.sin(angle)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    rot = torch.stack([c, s, z, -s, c, z, z, z, o], dim=1).view(-1, 3, 3)
    return points @ rot, rot


def _ortho6d(r6):
    """compute_rotation_matrix_from_ortho6d, VCN_VC.py:36-49 (normalize_vector clamps the norm at 1e-8, :12-22)"""
    def nrm(v):
        return v / torch.clamp(v.pow(2).sum(1).sqrt(), min=1e-8)[:, None]
    x = nrm(r6[:, 0:3])
    z = nrm(torch.cross(x, r6[:, 3:6], dim=1))
    y = torch.cross(z, x, dim=1)
    return torch.stack([x, y, z], dim=2)


class _Net:
    """Leaves (requires_grad) of a state_dict in the chosen dtype + the running statistics this forward would leave behind."""

    def __init__(self, sd, dtype):
        self.dtype, self.sd, self.leaves, self.buffers = dtype, sd, {}, {}

    def p(self, key):
        if key not in self.leaves:
            self.leaves[key] = torch.as_tensor(self.sd[key]).detach().to(self.dtype).clone().requires_grad_(True)
        return self.leaves[key]

    def lin(self, x, key):
        """Linear, or Conv1d(kernel 1) on channel-last rows: x (rows, K) W^T + b"""
        w = self.p(key + ".weight")
        return x @ w.reshape(w.shape[0], -1).t() + self.p(key + ".bias")

    def bn(self, x, key, eps=1e-5, momentum=0.1):
        """train-mode BatchNorm1d over the rows of x (rows, C)"""
        mean, var = x.mean(0), x.var(0, unbiased=False)
        rows = x.shape[0]
        rm, rv = (torch.as_tensor(self.sd[f"{key}.{k}"]).to(self.dtype) for k in ("running_mean", "running_var"))
        self.buffers[key + ".running_mean"] = (1 - momentum) * rm + momentum * mean.detach()
        self.buffers[key + ".running_var"] = (1 - momentum) * rv + momentum * var.detach() * rows / (rows - 1)
        self.buffers[key + ".num_batches_tracked"] = torch.as_tensor(self.sd[key + ".num_batches_tracked"]) + 1
        return (x - mean) / torch.sqrt(var + eps) * self.p(key + ".weight") + self.p(key + ".bias")


def _encoder(net, H, rows, n, prefix="encoder"):
    """FeatureEncoder.forward, VCN_VC.py:97-106, on channel-last rows (B n, 3) -> (B, 1024)"""
    f = H.act("enc.act1", net.bn(net.lin(rows, f"{prefix}.mlp_conv1.0"), f"{prefix}.mlp_conv1.1"), 0.0)
    f = net.lin(f, f"{prefix}.mlp_conv1.3")                                                 # (B n, 256)
    g = H.group_max("enc.max1", f, n)                                                       # :100
    f = torch.cat([g.repeat_interleave(n, dim=0), f], dim=1)                                # :101
    f = H.act("enc.act2", net.bn(net.lin(f, f"{prefix}.mlp_conv2.0"), f"{prefix}.mlp_conv2.1"), 0.0)
    return H.group_max("enc.max2", net.lin(f, f"{prefix}.mlp_conv2.3"), n)                  # :102-104


def _shape_fc(net, H, feat, nc=1024):
    h = H.act("shape_fc.act0", net.lin(feat, "shape_fc.0"), 0.0)
    h = H.act("shape_fc.act1", net.lin(h, "shape_fc.2"), 0.0)
    return net.lin(h, "shape_fc.4").reshape(-1, nc, 3)                                      # VCN_VC.py:204


def vcn_vc_train(sd, inp, dtype=torch.float64, hints=None, band=0.0):
    """VCN_VC.forward (VCN_VC.py:178-214), training mode.  -> (outputs with grad_fn, leaves {state_dict key: leaf}, buffers after the step,
    {decision: positions where the hint was followed against the oracle's own decision})"""
    net, H = _Net(sd, dtype), _Hints(hints, band)
    x = torch.as_tensor(inp).to(dtype)
    bs, n, _ = x.shape
    ang = torch.atan2(x[:, :, 1].mean(dim=1), x[:, :, 0].mean(dim=1))                       # :185
    fview, _ = _rotz(x, -ang)                                                               # :186
    mean = fview.mean(dim=1, keepdim=True)                                                  # :189
    h = H.act("pose.act0", net.lin((fview - mean).reshape(bs * n, 3), "pose_encoder.0"), 0.01)
    h = H.act("pose.act1", net.lin(h, "pose_encoder.2"), 0.01)
    pose_feat = H.group_max("pose.max", net.lin(h, "pose_encoder.4"), n)                    # :193
    rel = net.lin(H.act("pose_fc.act0", net.lin(pose_feat, "pose_fc.0"), 0.01), "pose_fc.2")  # :194
    centre = mean + rel[:, :3].unsqueeze(1)                                                 # :195-196
    rot = _ortho6d(rel[:, 3:9])                                                             # :197-198
    pc_cn = (fview - centre) @ rot.permute(0, 2, 1)                                         # :200
    feat = _encoder(net, H, pc_cn.reshape(bs * n, 3), n)                                    # :203
    coarse = _shape_fc(net, H, feat) @ rot + centre                                         # :204-205
    out, rz = _rotz(coarse, ang)                                                            # :208
    outs = {"coarse": out, "reg_rot": rot @ rz, "reg_centre": _rotz(centre, ang)[0].squeeze(1)}   # :211-212
    return outs, net.leaves, net.buffers, H.overridden


def vcn_cn_train(sd, inp, gt_boxes, dtype=torch.float64, hints=None, band=0.0):
    """VCN_CN.forward (VCN_CN.py:142-156 with transform.py:91-160), training mode; returns like vcn_vc_train"""
    net, H = _Net(sd, dtype), _Hints(hints, band)
    x, boxes = torch.as_tensor(inp).to(dtype), torch.as_tensor(gt_boxes).to(dtype)
    bs, n, _ = x.shape
    centre = boxes[:, :3].unsqueeze(1)
    pc = _rotz(x - centre, -boxes[:, 6])[0] / boxes[:, 3].view(-1, 1, 1)
    feat = _encoder(net, H, pc.reshape(bs * n, 3), n)
    coarse = _shape_fc(net, H, feat) * boxes[:, 3].view(-1, 1, 1)
    return {"coarse": _rotz(coarse, boxes[:, 6])[0] + centre}, net.leaves, net.buffers, H.overridden


def parity_loss(outs, up):
    """The scalar the parity tests differentiate: <coarse, up> + the sum of every other output"""
    return (outs["coarse"] * torch.as_tensor(up).to(outs["coarse"].dtype)).sum() + sum(v.sum() for k, v in outs.items() if k != "coarse")


def sample_index(numel, count=8192, seed=7):
    """The fixed entries of a flattened gradient the golden stores (all of them for tensors up to `count` entries)"""
    if numel <= count:
        return torch.arange(numel)
    return torch.randperm(numel, generator=torch.Generator().manual_seed(seed + numel))[:count].sort()[0]
