import math

import os

import torch
import torch.nn as nn

from . import functional as Fsp
from .core import SparseConvTensor
from .modules import SparseModule


def _triple(v):
    return [int(x) for x in v] if isinstance(v, (list, tuple)) else [int(v)] * 3


class SparseConvolution(SparseModule):
    """Base of SubMConv3d / SparseConv3d (the class pcdet/utils/spconv_utils.py:19 tests with isinstance).

    weight: (C_out, kz, ky, kx, C_in) — the spconv 2.x layout, so the reference's checkpoint loader
    (detector3d_template.py:330-359) adapts 1.x checkpoints to it and loads 2.x ones as they are."""

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None, **kwargs):
        super().__init__()
        assert ndim == 3 and groups == 1 and not transposed, "only 3-D submanifold and strided convs are built"
        assert not inverse, "inverse convolutions are not built"
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _triple(kernel_size), _triple(stride)
        self.padding, self.dilation = _triple(padding), _triple(dilation)
        self.subm, self.indice_key = subm, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, subm={self.subm}, indice_key={self.indice_key}")

    def weight_kio(self):
        """(K, C_in, C_out) view of the parameter (autograd-tracked)."""
        co = self.out_channels
        return self.weight.reshape(co, -1, self.in_channels).permute(1, 2, 0)

    def weight_kio_nograd(self):
        """The same view for consumers that only need its memory (fragment copies, launch lists): cached on the module while the parameter keeps
        its storage (two torch view ops per layer and call otherwise: 24 per chained step)."""
        w = self.weight
        hit = self.__dict__.get('_seevcn_kio')
        if hit is None or hit[0] is not w or hit[1] != w.data_ptr():
            with torch.no_grad():
                hit = (w, w.data_ptr(), w.reshape(self.out_channels, -1, self.in_channels).permute(1, 2, 0))     # ._base stays the Parameter: same fragment-cache key as weight_kio()
            self.__dict__['_seevcn_kio'] = hit
        return hit[2]

    def get_rulebook(self, x):
        rb = x.find_indice_pair(self.indice_key)
        if rb is not None and self.subm:
            assert rb.subm and rb.ksize == self.kernel_size, f"indice_key {self.indice_key} reused with a different kernel"
            return rb
        if rb is not None and not rb.subm and rb.in_indices is x.indices and rb.ksize == self.kernel_size:
            return rb                                   # built ahead of the layer loop by prebuild_rulebooks on these very indices
        if self.subm:
            rb = Fsp.build_subm_rulebook(x.indices, x.batch_size, x.spatial_shape, self.kernel_size, self.dilation)
        else:
            rb = Fsp.build_sparse_rulebook(x.indices, x.batch_size, x.spatial_shape, self.kernel_size, self.stride, self.padding,
                                           self.dilation)
        rb.in_indices, rb.in_shape = x.indices, list(x.spatial_shape)
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = rb
        return rb

    def forward(self, x):
        assert isinstance(x, SparseConvTensor)
        rb = self.get_rulebook(x)
        feats = Fsp.SparseConvFunction.apply(x.features, self.weight_kio(), rb)
        if self.bias is not None:
            feats = feats + self.bias
        out = SparseConvTensor(feats, rb.out_indices, rb.out_shape, x.batch_size, x.grid, x.indice_dict)
        return out

    def fusable_with(self, bn, x):
        """True when forward_bn_relu may replace self -> bn (-> ReLU): no bias, a training-mode nn.BatchNorm1d with
        affine parameters and running statistics on a channel count the fused kernels take, fp32 CUDA features, at least two output rows."""
        from . import norm
        return (self.bias is None and type(bn) is nn.BatchNorm1d and bn.training and bn.affine and bn.track_running_stats
                and bn.momentum is not None and norm.channels_fusable(self.out_channels) and x.features.is_cuda
                and x.features.dtype == torch.float32 and x.features.shape[0] > 1)

    def forward_bn_relu(self, x, bn, relu):
        """self -> bn (-> ReLU) as one autograd node (Fsp.SparseConvBNReLUFunction)."""
        rb = self.get_rulebook(x)
        if rb.n_out < 2:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {(rb.n_out, self.out_channels)}")
        feats = Fsp.SparseConvBNReLUFunction.apply(x.features, self.weight_kio(), rb, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                   bn.momentum, bn.eps, bool(relu), bn.num_batches_tracked)
        return SparseConvTensor(feats, rb.out_indices, rb.out_shape, x.batch_size, x.grid, x.indice_dict)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key, **kwargs)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, False,
                         indice_key=indice_key, **kwargs)


class SparseInverseConv3d(SparseConvolution):
    """Name kept so that `spconv.SparseInverseConv3d` resolves (spconv_backbone.py:16-18 builds it for the UNet backbones only, SURVEY.md 2:
    outside the hot path).  Constructing one fails at model-build time with a clear message."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("SparseInverseConv3d (UNet decoders of UNetV2 / PartA2) is outside the SEE-VCN hot path and not built; "
                                  "see INTEGRATION.md 'Unsupported surface'")


_registration_epoch = [0]


def _bump_epoch(module, name, submodule):
    _registration_epoch[0] += 1


# fires on every add_module / register_module / attribute assignment of a submodule, anywhere: the cached walks below are keyed on it
torch.nn.modules.module.register_module_module_registration_hook(_bump_epoch)


def _sparse_convs(root):
    """The SparseConvolution modules under `root` in definition order; the walk over nn.Module.modules() is cached on the root (0.25 ms per
    step otherwise) and redone whenever a submodule was registered or replaced anywhere since (global registration hook above) or the
    number of direct children changed (a deleted one)."""
    n = (_registration_epoch[0], len(root._modules))
    hit = root.__dict__.get('_seevcn_sparse_convs')
    if hit is None or hit[0] != n:
        hit = (n, [m for m in root.modules() if isinstance(m, SparseConvolution)])
        root.__dict__['_seevcn_sparse_convs'] = hit
    return hit[1]


# 0: rulebooks and plans built layer by layer (one device -> host read per strided level, ~57 launches per VoxelBackBone8x) instead of by
# Fsp.build_network_index (one read, 13 launches) -- A/B runs and tests; the tables are bit-identical either way
BATCH_INDEX = os.environ.get("SEEVCN_INDEX_BATCH", "1") != "0"
DEFER_INDEX_WORK = os.environ.get("SEEVCN_PREBUILD_DEFER", "1") != "0"     # layer-by-layer path only. 0: plans and submanifold rulebooks between the strided builders (A/B)


def _index_specs(root, x):
    """ConvSpec list of the keyed convolutions in front of `root`'s walk, or None when the batch form does not apply to them."""
    specs, seen, level = [], {}, 0
    for m in _sparse_convs(root):
        if m.indice_key is None:
            break
        if x.indice_dict.get(m.indice_key) is not None:
            return None                                   # something was built already: the layer-by-layer walk knows how to continue from it
        prev = seen.get(m.indice_key)
        if prev is not None and not (m.subm and prev[0].subm and prev[0].kernel_size == m.kernel_size and prev[0].dilation == m.dilation and prev[1] == level):
            return None                                   # a key shared by different kernels, by strided layers or across levels
        seen.setdefault(m.indice_key, (m, level))
        specs.append(Fsp.ConvSpec(m.indice_key, m.subm, m.kernel_size, m.stride, m.padding, m.dilation, m.in_channels, m.out_channels))
        if not m.subm:
            level += 1
    return specs or None


def prebuild_rulebooks(root, x, with_backward=True, n0_dev=None):
    """Build every rulebook (and conv plan) a network will need on `x` BEFORE its layers run.  A strided rulebook needs the number of output
    sites on the host (to allocate the next level), i.e. a device -> host sync; done lazily inside the layer loop, each of those syncs waits
    for all the convolution work queued so far and then leaves the GPU idle until the host has enqueued the next layers (spconv builds its
    indice pairs lazily, layer by layer: spconv_backbone.py:141-157 is the caller).  Default: Fsp.build_network_index -- the whole chain
    counted on the device, ONE read, all tables and plans in a handful of launches.  `root` is walked in definition order, which is the
    execution order of the reference's backbones; convolutions without an indice_key end the walk (they are then handled lazily).
    n0_dev: x.indices / x.features still have CAPACITY rows and the true count is this device word (a voxeliser called with sync=False): the
    count comes back with the same read and x is narrowed in place.  -> x"""
    if BATCH_INDEX or n0_dev is not None:
        specs = _index_specs(root, x) if BATCH_INDEX else None
        built = Fsp.build_network_index(x.indices, x.batch_size, x.spatial_shape, specs, n0_dev=n0_dev, with_backward=with_backward) if specs else None
        if built is not None:
            n0, rulebooks = built
            if n0_dev is not None:
                x.indices, x.features = x.indices[:n0], x.features[:n0]
            for key, rb in rulebooks.items():
                if rb.in_indices.data_ptr() == x.indices.data_ptr() and rb.in_indices.shape == x.indices.shape:
                    rb.in_indices = x.indices         # level 0: the very tensor the layers will present (get_rulebook compares identities)
                if rb.subm and rb.out_indices.data_ptr() == x.indices.data_ptr():
                    rb.out_indices = x.indices
                x.indice_dict[key] = rb
            if with_backward:                         # the weight gradients' equal-pieces plans, one per table, all in two launches behind the tables
                Fsp.build_wgrad_plans([(rulebooks[sp.key], sp.cin, sp.cout) for sp in specs])
            return x
        if n0_dev is not None:
            from .. import _lib
            n0 = _lib.host_int(n0_dev)
            x.indices, x.features = x.indices[:n0], x.features[:n0]
    _prebuild_layer_by_layer(root, x, with_backward)
    return x


def _prebuild_layer_by_layer(root, x, with_backward):
    idx, shape = x.indices, list(x.spatial_shape)
    # Pass 1: only what the device -> host reads hang on -- the STRIDED rulebooks, each needing the output sites of the one before.  Everything
    # else (submanifold rulebooks, all plans) is enqueued in pass 2, behind the last read: queued between the strided builders (round 2) it sat
    # in front of every later read -- ~145 us of index kernels per level that the host waited for four times per step.
    todo = []                                             # (module, its input indices, their shape) in execution order
    for m in _sparse_convs(root):
        if m.indice_key is None:
            break
        if m.subm and DEFER_INDEX_WORK:
            todo.append((m, idx, list(shape)))
            continue
        rb = x.indice_dict.get(m.indice_key)
        if rb is None:
            if m.subm:
                rb = Fsp.build_subm_rulebook(idx, x.batch_size, shape, m.kernel_size, m.dilation)
            else:
                rb = Fsp.build_sparse_rulebook(idx, x.batch_size, shape, m.kernel_size, m.stride, m.padding, m.dilation)
            rb.in_indices, rb.in_shape = idx, list(shape)
            x.indice_dict[m.indice_key] = rb
        elif rb.in_indices is not idx:
            break                                         # the key is bound to other indices: not the simple chain this walk assumes
        if DEFER_INDEX_WORK:
            todo.append((m, idx, list(shape)))
        else:
            rb.plan("fwd", m.in_channels, m.out_channels)
            if with_backward:
                rb.plan("bwd", m.out_channels, m.in_channels)
                rb.wgrad_plan(m.in_channels, m.out_channels)
        if not m.subm:
            idx, shape = rb.out_indices, list(rb.out_shape)
    for m, idx, shape in todo:                            # pass 2: no host reads from here on
        rb = x.indice_dict.get(m.indice_key)
        if rb is None:
            rb = Fsp.build_subm_rulebook(idx, x.batch_size, shape, m.kernel_size, m.dilation)
            rb.in_indices, rb.in_shape = idx, list(shape)
            x.indice_dict[m.indice_key] = rb
        elif rb.in_indices is not idx:
            return
        rb.plan("fwd", m.in_channels, m.out_channels)
        if with_backward:
            rb.plan("bwd", m.out_channels, m.in_channels)
            rb.wgrad_plan(m.in_channels, m.out_channels)


def refresh_weight_fragments(root):
    """One launch that re-lays the MFMA fragment copies of every planned-kernel convolution under `root` (Fsp.fragment_cache.refresh_all)."""
    if not Fsp.USE_PLAN:
        return
    ws = [m.weight_kio_nograd() for m in _sparse_convs(root) if m.weight.is_cuda]
    with torch.no_grad():
        Fsp.fragment_cache.refresh_all(ws)
