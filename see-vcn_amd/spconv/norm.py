"""Fused BatchNorm1d(+ReLU) over voxel feature matrices, used by SparseSequential / SparseBasicBlock in place of the separate
torch kernels for the `norm_fn -> ReLU` tail of the reference's post_act_block (spconv_backbone.py:9-27).  The modules stay
plain `nn.BatchNorm1d` / `nn.ReLU` (same state_dict keys, same running-statistics semantics); only the arithmetic moves."""
import os

import torch
import torch.nn as nn

from .. import _lib


_SCRATCH_BYTES = {}


def _scratch(channels, device):
    n = _SCRATCH_BYTES.get(channels)
    if n is None:
        n = _SCRATCH_BYTES[channels] = _lib.load().sv_batchnorm_scratch_bytes(channels)
    return _lib.workspace.scratch(f"bn{channels}", n, device)


STATS_IN_CONV = os.environ.get("SEEVCN_BN_STATS_IN_CONV", "1") != "0"      # 0: the fused conv + BatchNorm node runs the separate statistics pass (A/B runs)


def partial_address(channels, device):
    """Device address where the producer of a BatchNorm input leaves its partial sums: behind the 4 * C coefficient floats of the norm's scratch."""
    return _scratch(channels, device).data_ptr() + 16 * channels


def bn_forward_raw(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, num_batches_tracked=None, n_partials=0):
    """y, save_mean, save_invstd (the last two None in eval mode) of the fused BatchNorm(+ReLU) forward on a contiguous (N, C) matrix.  n_partials > 0:
    the statistics' first pass is already in the scratch (partial_address), written by the kernel that produced x."""
    lib = _lib.load()
    n, c = x.shape
    y = torch.empty_like(x)
    if training:
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    else:
        mean = invstd = None
    if training and n_partials:
        _lib.check(lib.sv_batchnorm_relu_forward_partial(_lib.ptr(x), n, c, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean), _lib.ptr(running_var),
                                                         float(momentum), float(eps), int(relu), _lib.ptr(_scratch(c, x.device)), int(n_partials), _lib.ptr(y),
                                                         _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(num_batches_tracked), _lib.stream()),
                   "sv_batchnorm_relu_forward_partial")
        return y, mean, invstd
    _lib.check(lib.sv_batchnorm_relu_forward(_lib.ptr(x), n, c, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean), _lib.ptr(running_var),
                                             float(momentum), float(eps), int(training), int(relu), _lib.ptr(_scratch(c, x.device)), _lib.ptr(y),
                                             _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(num_batches_tracked), _lib.stream()), "sv_batchnorm_relu_forward")
    return y, mean, invstd


def bn_backward_raw(x, dy, gamma, beta, mean, invstd, relu):
    """dx, dgamma, dbeta of the training-mode fused BatchNorm(+ReLU); x is the BatchNorm INPUT (the ReLU mask is recomputed from it)."""
    lib = _lib.load()
    n, c = x.shape
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    _lib.check(lib.sv_batchnorm_relu_backward(_lib.ptr(x), _lib.ptr(dy), n, c, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(mean), _lib.ptr(invstd),
                                              int(relu), _lib.ptr(_scratch(c, x.device)), _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta),
                                              _lib.stream()), "sv_batchnorm_relu_backward")
    return dx, dgamma, dbeta


class _BatchNormReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, num_batches_tracked=None):
        x = x.contiguous()
        y, mean, invstd = bn_forward_raw(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, num_batches_tracked)
        ctx.relu, ctx.training = relu, training
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        assert ctx.training, "fused BatchNorm backward is only defined for training mode"
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        dx, dgamma, dbeta = bn_backward_raw(x, dy.contiguous(), gamma, beta, mean, invstd, ctx.relu)
        return dx, (dgamma if gamma is not None else None), (dbeta if beta is not None else None), None, None, None, None, None, None, None


def channels_fusable(c):
    return 4 <= c <= 512 and c % 4 == 0 and 256 % (c // 4) == 0


def fusable(bn, x):
    c = x.shape[1] if x.dim() == 2 else 0
    return (isinstance(bn, nn.BatchNorm1d) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0
            and c >= 4 and c <= 512 and c % 4 == 0 and 256 % (c // 4) == 0 and bn.momentum is not None
            and (bn.training or bn.track_running_stats) and (not x.requires_grad or bn.training))


def batch_norm_relu(bn, x, relu):
    """y = [relu](bn(x)) with bn an nn.BatchNorm1d (training or eval), x (N,C) float32 CUDA."""
    training = bn.training or not bn.track_running_stats
    if training and x.shape[0] == 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
    nbt = bn.num_batches_tracked if (training and bn.track_running_stats) else None   # incremented inside the kernel chain
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BatchNormReLU.apply(x, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps, training, relu, nbt)
