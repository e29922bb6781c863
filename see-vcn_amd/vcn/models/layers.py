"""Parameter containers with the reference's state_dict keys + the HIP execution of the VCN layer chain."""
import torch
import torch.nn as nn

from ... import _lib

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
LRELU_SLOPE = 0.01  # nn.LeakyReLU() default, VCN_VC.py:118-120,126


def fc_layers(layer_dims, last_as_linear=True):
    """Linear/ReLU stack with the reference's module indices (VCN_VC.py:66-79) so state_dict keys match."""
    layers, in_ch = [], layer_dims[0]
    for out_ch in layer_dims[1:]:
        if out_ch == layer_dims[-1] and last_as_linear:
            layers.append(nn.Linear(in_ch, out_ch))
            break
        layers += [nn.Linear(in_ch, out_ch), nn.ReLU(inplace=True)]
        in_ch = out_ch
    return nn.Sequential(*layers)


class FeatureEncoder(nn.Module):
    """Holds the parameters of the reference FeatureEncoder (VCN_VC.py:81-106); executed by `encode()`."""

    def __init__(self, dims):
        super().__init__()
        self.mlp_conv1 = nn.Sequential(
            nn.Conv1d(dims[0], dims[1], 1), nn.BatchNorm1d(dims[1]), nn.ReLU(inplace=True), nn.Conv1d(dims[1], dims[2], 1))
        self.mlp_conv2 = nn.Sequential(
            nn.Conv1d(dims[3], dims[4], 1), nn.BatchNorm1d(dims[4]), nn.ReLU(inplace=True), nn.Conv1d(dims[4], dims[5], 1))


def fold_conv_bn(conv, bn):
    """Eval-mode BatchNorm folded into the preceding 1x1 conv: w' = w*g/sqrt(var+eps), b' = (b-mean)*g/sqrt(var+eps)+beta."""
    w = conv.weight.detach().reshape(conv.out_channels, -1).float()
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(conv.out_channels, device=w.device)
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    return (w * scale[:, None]).contiguous(), ((b - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()).contiguous()


def conv_wb(conv):
    w = conv.weight.detach().reshape(conv.out_channels, -1).float().contiguous()
    b = conv.bias.detach().float().contiguous() if conv.bias is not None else None
    return w, b


def lin_wb(lin):
    return lin.weight.detach().float().contiguous(), (lin.bias.detach().float().contiguous() if lin.bias is not None else None)


class PreparedCache:
    """Folded / contiguous weights cached until any parameter or buffer changes (tensor version counters)."""

    def __init__(self, module, builder):
        self.module, self.builder, self.key, self.value = module, builder, None, None

    def invalidate(self):
        """Version counters miss updates made by torch's fused optimisers (SGD / Adam with fused=True do not bump them): the owning modules call
        this on every train() / eval() switch, so weights trained with such an optimiser are re-folded before inference."""
        self.key = None

    def get(self):
        tensors = list(self.module.parameters()) + list(self.module.buffers())
        key = tuple((t.data_ptr(), t._version, t.device) for t in tensors)
        if key != self.key:
            self.value = self.builder()
            self.key = key
        return self.value


def gemm(a, w, bias=None, act=ACT_NONE, group_bias=None, rows_per_group=1, store=True, group_max=None, slope=LRELU_SLOPE, row_group=None, m_dev=None, tag=None):
    """act(a @ w.T + bias + group_bias[group(row)]) via sv_gemm_bias_act (fp32 MFMA); group(row) = row // rows_per_group, or
    row_group[row] (int32, non-decreasing) for ragged groups.  m_dev (with row_group): the number of valid rows of `a` lives on the device
    (int32 tensor); a.shape[0] is then the capacity, the output keeps capacity rows and only the first *m_dev are computed.

    Returns the (M,N) output (or None when store=False); `group_max` (groups, N) must be pre-filled with -inf."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.is_contiguous() and w.is_contiguous()
    if store and m_dev is not None and tag is not None:
        # capacity-sized intermediate of the lazy-row forward (M = B n rows addressable, ~U computed): a grow-only per-stream buffer instead of a
        # fresh 16-134 MB allocation per layer and call (the caching allocator went back to hipMalloc for them after every empty_cache())
        out = _lib.workspace.scratch(f"vcn_{tag}", M * N * 4, a.device)[:M * N * 4].view(torch.float32).view(M, N)
    else:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device) if store else None
    if row_group is None:
        assert m_dev is None
        rc = lib.sv_gemm_bias_act(_lib.ptr(a), K, _lib.ptr(w), K, _lib.ptr(bias), _lib.ptr(group_bias), int(rows_per_group),
                                  _lib.ptr(out), N, _lib.ptr(group_max), M, N, K, int(act), float(slope), _lib.stream())
    elif m_dev is None:
        assert row_group.dtype == torch.int32 and row_group.shape[0] == M
        rc = lib.sv_gemm_bias_act_ragged(_lib.ptr(a), K, _lib.ptr(w), K, _lib.ptr(bias), _lib.ptr(group_bias), _lib.ptr(row_group),
                                         _lib.ptr(out), N, _lib.ptr(group_max), M, N, K, int(act), float(slope), _lib.stream())
    else:
        assert row_group.dtype == torch.int32 and row_group.shape[0] == M and m_dev.dtype == torch.int32
        rc = lib.sv_gemm_bias_act_ragged_dev(_lib.ptr(a), K, _lib.ptr(w), K, _lib.ptr(bias), _lib.ptr(group_bias), _lib.ptr(row_group),
                                             _lib.ptr(out), N, _lib.ptr(group_max), M, _lib.ptr(m_dev), N, K, int(act), float(slope), _lib.stream())
    _lib.check(rc, "sv_gemm_bias_act")
    return out


def distinct_rows(x, sync=True):
    """x (B,n,3) -> (sel (U,) int64 flat row indices into x.view(B*n,3), row_group (U,) int32 object of each kept row).

    ResamplePoints (vcn/datasets/data_transforms.py:254-262) tiles an object's Ni points to n = 1024, so a cloud holds only
    Ni distinct rows; the per-point layers (Conv1d k=1, eval-mode BatchNorm folded) and the max-pools over points give
    bit-identical results on the distinct rows alone.  sync=True: one host sync (the number of kept rows).  sync=False: no read --
    (sel, row_group) keep their B*n capacity rows and the third value is U as a device int32 tensor for the *_dev / *_gather layer entries."""
    lib = _lib.load()
    B, n, _ = x.shape
    idx = torch.empty((B, n), dtype=torch.int32, device=x.device)
    cnt = torch.empty((B,), dtype=torch.int32, device=x.device)
    _lib.check(lib.sv_unique_rows(_lib.ptr(x), B, n, _lib.ptr(idx), _lib.ptr(cnt), _lib.stream()), "sv_unique_rows")
    sel = torch.empty((B * n,), dtype=torch.int64, device=x.device)
    row_group = torch.empty((B * n,), dtype=torch.int32, device=x.device)
    total = torch.empty((1,), dtype=torch.int32, device=x.device)
    _lib.check(lib.sv_unique_rows_compact(_lib.ptr(idx), _lib.ptr(cnt), B, n, _lib.ptr(sel), _lib.ptr(row_group), _lib.ptr(total), _lib.stream()),
               "sv_unique_rows_compact")
    if not sync:
        return sel, row_group, total
    u = _lib.host_int(total)                                                                                     # sync: U rows
    sel, row_group = sel[:u], row_group[:u]
    return sel, row_group


def pointwise3(xyz, w, b, act, slope=LRELU_SLOPE, sel=None, m_dev=None, tag=None):
    """K = 3 first layer on the rows of xyz (M, 3) -- or, with sel (capacity,) int64 and m_dev, on xyz[sel[m]] for m < *m_dev (the output keeps
    the capacity's rows; gather and count stay on the device)."""
    lib = _lib.load()
    C = w.shape[0]
    if sel is not None and m_dev is not None:
        M = sel.shape[0]
        out = (_lib.workspace.scratch(f"vcn_{tag}", M * C * 4, xyz.device)[:M * C * 4].view(torch.float32).view(M, C) if tag is not None
               else torch.empty((M, C), dtype=torch.float32, device=xyz.device))
        rc = lib.sv_pointwise_conv3_gather(_lib.ptr(xyz), _lib.ptr(sel), M, _lib.ptr(m_dev), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), C, int(act), float(slope),
                                           _lib.stream())
        _lib.check(rc, "sv_pointwise_conv3_gather")
        return out
    if sel is not None:
        xyz = xyz[sel]
    M = xyz.shape[0]
    out = torch.empty((M, C), dtype=torch.float32, device=xyz.device)
    rc = lib.sv_pointwise_conv3(_lib.ptr(xyz), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, C, int(act), float(slope), _lib.stream())
    _lib.check(rc, "sv_pointwise_conv3")
    return out


def neg_inf(shape, device):
    lib = _lib.load()
    t = torch.empty(shape, dtype=torch.float32, device=device)
    _lib.check(lib.sv_fill_f32(_lib.ptr(t), t.numel(), float("-inf"), _lib.stream()), "sv_fill_f32")
    return t


class NegInfPool:
    """The -inf start values of a forward's column-max outputs from ONE allocation and ONE fill launch (three 5 us launches in VCN_VC's forward
    otherwise); take() hands out consecutive (rows, cols) views."""

    def __init__(self, numel, device):
        self.buf, self.used = neg_inf((int(numel),), device), 0

    def take(self, shape):
        n = int(shape[0]) * int(shape[1])
        assert self.used + n <= self.buf.numel()
        out = self.buf[self.used:self.used + n].view(shape)
        self.used += n
        return out


def prepare_encoder(enc):
    """Weights of FeatureEncoder in execution form. mlp_conv2[0] is split into its global (first 256 inputs)
    and local halves: the global half multiplies a per-object constant, so it becomes a per-object bias."""
    w1a, b1a = fold_conv_bn(enc.mlp_conv1[0], enc.mlp_conv1[1])
    w1b, b1b = conv_wb(enc.mlp_conv1[3])
    w2a, b2a = fold_conv_bn(enc.mlp_conv2[0], enc.mlp_conv2[1])
    w2b, b2b = conv_wb(enc.mlp_conv2[3])
    cg = w1b.shape[0]  # channels of the global feature (256)
    return dict(w1a=w1a, b1a=b1a, w1b=w1b, b1b=b1b, w2a_g=w2a[:, :cg].contiguous(), w2a_l=w2a[:, cg:].contiguous(),
                b2a=b2a, w2b=w2b, b2b=b2b)


def encode(p, pts, batch, n, row_group=None, sel=None, m_dev=None, pool=None):
    """FeatureEncoder.forward (VCN_VC.py:97-106) on channel-last activations. pts: (B*n, 3); with row_group only the distinct rows
    pts[sel] are run (row_group = their objects; m_dev: their number on the device, sel / row_group at capacity) -> (B, 1024)."""
    dev = pts.device
    f1 = pointwise3(pts, p["w1a"], p["b1a"], ACT_RELU, sel=sel, m_dev=m_dev, tag="enc_f1")         # conv 3->128 + BN + ReLU
    g1 = pool.take((batch, p["w1b"].shape[0])) if pool is not None else neg_inf((batch, p["w1b"].shape[0]), dev)
    local = gemm(f1, p["w1b"], p["b1b"], ACT_NONE, rows_per_group=n, group_max=g1, row_group=row_group, m_dev=m_dev, tag="enc_local")  # conv 128->256, max over n
    gb = gemm(g1, p["w2a_g"], None, ACT_NONE)                                        # global half of conv 512->512
    f2 = gemm(local, p["w2a_l"], p["b2a"], ACT_RELU, group_bias=gb, rows_per_group=n, row_group=row_group, m_dev=m_dev, tag="enc_f2")  # + BN + ReLU
    g2 = pool.take((batch, p["w2b"].shape[0])) if pool is not None else neg_inf((batch, p["w2b"].shape[0]), dev)
    gemm(f2, p["w2b"], p["b2b"], ACT_NONE, rows_per_group=n, store=False, group_max=g2, row_group=row_group, m_dev=m_dev)  # conv 512->1024, max over n
    return g2


def prepare_fc(seq):
    return [lin_wb(m) for m in seq if isinstance(m, nn.Linear)]


def run_fc(layers, x, hidden_act):
    for i, (w, b) in enumerate(layers):
        x = gemm(x, w, b, hidden_act if i + 1 < len(layers) else ACT_NONE)
    return x


# ---------------------------------------------------------------------------------------------------------------- training mode, own kernels
TAPS = None   # tests set this to a dict: the training forward then leaves the tensors its discrete decisions are taken on (activation outputs, pool
              # inputs / outputs) under the decision names of oracle/vcn_train.py, so that the float64 oracle can follow the device's branch inside its band


def _tap(name, *tensors):
    if TAPS is not None:
        TAPS[name] = tuple(t.detach() for t in tensors) if len(tensors) > 1 else tensors[0].detach()


def _bn_relu(bn, x):
    """BatchNorm1d + ReLU: the fused kernels when the norm qualifies (spconv.norm.fusable), the module + torch.relu otherwise (a norm frozen in eval()
    inside a training model, channel counts the kernels do not take)"""
    from ...spconv import norm
    if norm.fusable(bn, x):
        return norm.batch_norm_relu(bn, x, True)
    return torch.relu(bn(x))


def encode_train(enc, pts_rows, batch, n):
    """FeatureEncoder.forward (VCN_VC.py:97-106) in TRAINING mode on channel-last rows (B n, 3) -> (B, 1024), differentiable: Conv1d(k=1) =
    dense_ops.linear (fp32 MFMA forward, hand-written backward), BatchNorm1d(batch statistics) + ReLU = the fused kernels of spconv.norm, max over
    the points = dense_ops.segment_max.  The conv over cat([global.expand, local]) is local W_l^T + (global W_g^T)[object] + b."""
    from ... import dense_ops as D
    c1, c2 = enc.mlp_conv1, enc.mlp_conv2
    f = _bn_relu(c1[1], D.linear(pts_rows, c1[0].weight.squeeze(-1), c1[0].bias))
    _tap("enc.act1", f)
    local = D.linear(f, c1[3].weight.squeeze(-1), c1[3].bias)                             # (B n, 256)
    g1 = D.segment_max(local, n)                                                          # (B, 256)
    _tap("enc.max1", local, g1)
    cg = g1.shape[1]
    w2 = c2[0].weight.squeeze(-1)
    gb = D.linear(g1, w2[:, :cg], None)                                                   # the global half: one row per object
    f2 = _bn_relu(c2[1], D.linear(local, w2[:, cg:], c2[0].bias, group_bias=gb, rows_per_group=n))
    _tap("enc.act2", f2)
    z = D.linear(f2, c2[3].weight.squeeze(-1), c2[3].bias)
    g2 = D.segment_max(z, n)                                                              # (B, 1024)
    _tap("enc.max2", z, g2)
    return g2


def run_fc_train(seq, x, tap=None):
    """An nn.Sequential of Linear [+ ReLU / LeakyReLU] modules on dense_ops.linear (activation fused into the layer in front of it)."""
    from ... import dense_ops as D
    mods = list(seq)
    i = j = 0
    while i < len(mods):
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        act = D.ACT_RELU if isinstance(nxt, nn.ReLU) else D.ACT_LRELU if isinstance(nxt, nn.LeakyReLU) else D.ACT_NONE
        x = D.linear(x, mods[i].weight, mods[i].bias, act, nxt.negative_slope if act == D.ACT_LRELU else 0.0)
        if act != D.ACT_NONE and tap is not None:
            _tap(f"{tap}.act{j}", x)
            j += 1
        i += 2 if act != D.ACT_NONE else 1
    return x
