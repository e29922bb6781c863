"""Autograd functions over the library's dense-layer kernels: Linear / Conv1d(k=1) on channel-last rows with a hand-written backward, the max over
an object's points, and the broadcast of a per-object feature into a per-point layer.  They carry the TRAINING forward / backward of the layer
stacks the reference runs through cuDNN / cuBLAS under autograd -- VCN_VC / VCN_CN in training mode (see/surface_completion/models/vcn/models/
VCN_VC.py:97-106,116-131,178-214) and PV-RCNN's point head, feature fusion and RoI-head FC stacks (detector3d/pcdet/models/dense_heads/
point_head_simple.py, backbones_3d/pfe/voxel_set_abstraction.py:168-172, roi_heads/pvrcnn_head.py:171-176):

  linear(x, weight, bias, act, slope, group_bias, rows_per_group)   y = act(x W^T + b [+ group_bias[row // rows_per_group]])
      forward  sv_gemm_bias_act (fp32 MFMA; K = 3: sv_pointwise_conv3; odd shapes: sv_gemm_strided)
      backward dz = dy act'(y) (sv_act_backward), dX = dz W (sv_gemm_bias_act on W^T, or sv_gemm_strided), dW = dz^T X (sv_gemm_tn: contraction
               over the rows, no transposed copy of an activation), db = column sums, d group_bias = sv_segment_sum
  segment_max(x, rows_per_group)                                     max over each group's rows, gradient to the arg-max rows
BatchNorm1d in training mode is seevcn_amd.spconv.norm.batch_norm_relu (the fused kernels of the sparse backbone work on any (rows, C) matrix).
There is no CPU fallback."""
import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2


def _scratch(name, nbytes, device):
    return _lib.workspace.scratch(name, nbytes, device)


def _gemm_nt(a, w, bias, act, slope, group_bias=None, rows_per_group=1):
    """act(a (M,K) @ w (N,K)^T + bias + group_bias[row // rows_per_group]) on the kernels of the eval path; -> (M, N)"""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if M == 0:
        return out
    if K == 3 and group_bias is None and N % 4 == 0:
        _lib.check(lib.sv_pointwise_conv3(_lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out), M, N, int(act), float(slope), _lib.stream()), "sv_pointwise_conv3")
        return out
    if K % 32 == 0 and lib.sv_gemm_splitk_splits(M, N, K) > 1:             # few output tiles, long K (the 27 648 -> 256 layer of PV-RCNN's RoI head)
        sc = _scratch("dense_splitk", lib.sv_gemm_splitk_scratch_bytes(M, N, K), a.device)
        _lib.check(lib.sv_gemm_bias_act_splitk(_lib.ptr(a), K, _lib.ptr(w), K, _lib.ptr(bias), _lib.ptr(group_bias), int(rows_per_group), _lib.ptr(out), N, M, N, K,
                                               int(act), float(slope), _lib.ptr(sc), _lib.stream()), "sv_gemm_bias_act_splitk")
        return out
    if K % 32 == 0:
        _lib.check(lib.sv_gemm_bias_act(_lib.ptr(a), K, _lib.ptr(w), K, _lib.ptr(bias), _lib.ptr(group_bias), int(rows_per_group), _lib.ptr(out), N, None, M, N, K,
                                        int(act), float(slope), _lib.stream()), "sv_gemm_bias_act")
        return out
    sc = _scratch("dense_sg", lib.sv_gemm_strided_scratch_bytes(M, N, K), a.device)
    _lib.check(lib.sv_gemm_strided(_lib.ptr(a), K, 1, _lib.ptr(w), 1, K, _lib.ptr(out), N, M, N, K, _lib.ptr(sc), _lib.stream()), "sv_gemm_strided")
    if bias is not None:
        out += bias
    if group_bias is not None:
        out += group_bias.repeat_interleave(rows_per_group, dim=0)
    if act != ACT_NONE:
        out = torch.where(out > 0, out, out * (slope if act == ACT_LRELU else 0.0))
    return out


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, slope, group_bias, rows_per_group):
        _lib.require_cuda(x, weight)
        x = x.contiguous().float()
        w = weight.contiguous().float()
        b = None if bias is None else bias.contiguous().float()
        gb = None if group_bias is None else group_bias.contiguous().float()
        y = _gemm_nt(x, w, b, act, slope, gb, rows_per_group)
        ctx.act, ctx.slope, ctx.rows_per_group = int(act), float(slope), int(rows_per_group)
        ctx.has_bias, ctx.has_gb = bias is not None, group_bias is not None
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, y = ctx.saved_tensors
        M, K = x.shape
        N = w.shape[0]
        dev = x.device
        dz = dy.contiguous().float()
        if ctx.act != ACT_NONE:
            out = torch.empty_like(dz)
            _lib.check(lib.sv_act_backward(_lib.ptr(dz), _lib.ptr(y), dz.numel(), ctx.act, ctx.slope, _lib.ptr(out), _lib.stream()), "sv_act_backward")
            dz = out
        dx = dw = db = dgb = None
        if ctx.needs_input_grad[0]:
            if N % 32 == 0 and M > 0:
                wt = w.t().contiguous()                               # (K, N): the data gradient is the same NT kernel on the transposed weight
                dx = _gemm_nt(dz, wt, None, ACT_NONE, 0.0)
            else:
                dx = torch.empty((M, K), dtype=torch.float32, device=dev)
                sc = _scratch("dense_sg", lib.sv_gemm_strided_scratch_bytes(M, K, N), dev)
                _lib.check(lib.sv_gemm_strided(_lib.ptr(dz), N, 1, _lib.ptr(w), K, 1, _lib.ptr(dx), K, M, K, N, _lib.ptr(sc), _lib.stream()), "sv_gemm_strided")
        if ctx.needs_input_grad[1]:
            dw = torch.empty((N, K), dtype=torch.float32, device=dev)
            sc = _scratch("dense_tn", lib.sv_gemm_tn_scratch_bytes(M, N, K), dev)
            _lib.check(lib.sv_gemm_tn(_lib.ptr(dz), N, _lib.ptr(x), K, _lib.ptr(dw), K, M, N, K, _lib.ptr(sc), _lib.stream()), "sv_gemm_tn")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty((N,), dtype=torch.float32, device=dev)
            sc = _scratch("dense_cs", lib.sv_column_sums_scratch_bytes(M, N), dev)
            _lib.check(lib.sv_column_sums(_lib.ptr(dz), N, M, N, _lib.ptr(db), _lib.ptr(sc), _lib.stream()), "sv_column_sums")
        if ctx.has_gb and ctx.needs_input_grad[5]:
            groups = M // ctx.rows_per_group
            dgb = torch.empty((groups, N), dtype=torch.float32, device=dev)
            _lib.check(lib.sv_segment_sum(_lib.ptr(dz), N, groups, ctx.rows_per_group, N, _lib.ptr(dgb), _lib.stream()), "sv_segment_sum")
        return dx, dw, db, None, None, dgb, None


def linear(x, weight, bias=None, act=ACT_NONE, slope=0.01, group_bias=None, rows_per_group=1):
    """x (M, K) rows, weight (N, K) [a Conv1d(k=1) weight squeezed], bias (N) -> act(x W^T + b + group_bias[row // rows_per_group]) (M, N)."""
    assert x.dim() == 2 and weight.dim() == 2 and x.shape[1] == weight.shape[1]
    if group_bias is not None:
        assert x.shape[0] % rows_per_group == 0 and group_bias.shape == (x.shape[0] // rows_per_group, weight.shape[0])
    return _Linear.apply(x, weight, bias, act, slope, group_bias, rows_per_group)


class _SegmentMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rows_per_group):
        lib = _lib.load()
        _lib.require_cuda(x)
        x = x.contiguous().float()
        M, C = x.shape
        groups = M // rows_per_group
        out = torch.empty((groups, C), dtype=torch.float32, device=x.device)
        arg = torch.empty((groups, C), dtype=torch.int32, device=x.device)
        _lib.check(lib.sv_segment_max(_lib.ptr(x), C, groups, int(rows_per_group), C, _lib.ptr(out), _lib.ptr(arg), _lib.stream()), "sv_segment_max")
        ctx.rows_per_group, ctx.shape = int(rows_per_group), (M, C)
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable()
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (arg,) = ctx.saved_tensors
        M, C = ctx.shape
        dx = torch.empty((M, C), dtype=torch.float32, device=dout.device)
        _lib.check(lib.sv_segment_max_backward(_lib.ptr(dout.contiguous().float()), _lib.ptr(arg), M // ctx.rows_per_group, ctx.rows_per_group, C, _lib.ptr(dx), C,
                                               _lib.stream()), "sv_segment_max_backward")
        return dx, None


def segment_max(x, rows_per_group):
    """x (groups * rows_per_group, C) -> (groups, C): torch.max(x.view(groups, rows_per_group, C), dim=1)[0] with its gradient."""
    assert x.dim() == 2 and x.shape[0] % rows_per_group == 0
    return _SegmentMax.apply(x, int(rows_per_group))


def _has_hooks(m):
    return bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None) or getattr(m, "_forward_hooks_with_kwargs", None))


def run_sequential(layers, x):
    """An nn.Sequential of Linear | Conv1d(kernel 1) | BatchNorm1d | ReLU | Dropout modules on (rows, C) CUDA fp32 features through this
    library's kernels -- linear() for the products (hand-written backward), the fused BatchNorm(+ReLU) kernels of spconv.norm for the norms (batch
    statistics in training, running statistics in eval); modules it does not know, and anything not CUDA fp32, run as they are.  The parameters,
    buffers and state_dict keys stay the modules' own: PV-RCNN's point head, feature fusion and RoI-head FC stacks (point_head_template.py:20-32,
    voxel_set_abstraction.py:168-172, roi_head_template.py:27-38 / pvrcnn_head.py:171-176) call this instead of `layers(x)`."""
    import torch.nn as nn
    from .spconv import norm
    mods = list(layers)
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2) or _has_hooks(layers) or any(_has_hooks(m) for m in mods):
        return layers(x)                                              # hooks observe module calls: the fused walk below calls no module (spconv/chain.py does the same)
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if isinstance(m, nn.Linear) or (isinstance(m, nn.Conv1d) and m.kernel_size == (1,) and m.stride == (1,) and m.padding == (0,) and m.groups == 1):
            w = m.weight if m.weight.dim() == 2 else m.weight.squeeze(-1)
            fuse_relu = isinstance(nxt, nn.ReLU)
            x = linear(x, w, m.bias, ACT_RELU if fuse_relu else ACT_NONE)
            i += 2 if fuse_relu else 1
        elif isinstance(m, nn.BatchNorm1d) and norm.fusable(m, x):
            relu = isinstance(nxt, nn.ReLU)
            x = norm.batch_norm_relu(m, x, relu)
            i += 2 if relu else 1
        else:
            x = m(x)
            i += 1
    return x
