"""A chain of [sparse conv -> BatchNorm1d -> ReLU] blocks as ONE autograd node over the launch-list executor (sv_run_ops, csrc/sequencer.hip).

The reference's VoxelBackBone8x.forward (detector3d/pcdet/models/backbones_3d/spconv_backbone.py:128-180) is 13 such blocks; through the module
tree each block costs the host an autograd node, three output allocations and two or three ctypes calls per direction (2.2 ms of Python per
step for 4.0 ms of GPU time on the bench workload).  Here the forward of the whole chain is written as one list of operations -- every
argument is known before the first kernel runs: the rulebooks and plans are built ahead, the activations are slices of one allocation -- and
enqueued with one call; the backward (BatchNorm backward, weight gradient, data gradient per block, last to first) likewise.  Same kernels,
same arithmetic, same parameters and running statistics as the per-module path (SparseSequential), which stays the fallback for anything
this does not take (eval mode, hooks, a conv bias, a layer without a plan for its data gradient, residual blocks)."""
import os
import struct

import numpy as np
import torch

from .. import _lib
from . import functional as Fsp
from . import norm
from .core import SparseConvTensor

CHAIN_OFF = os.environ.get("SEEVCN_CHAIN", "1") == "0"          # 0: every block through its own modules (A/B runs, tests)
# 1 (default since round 5): a BatchNorm backward takes its two per-channel sums from the epilogue of the data-gradient launch above it
# (sv_sparse_conv_dgrad_planned_bn) instead of reducing them in a pass of its own.  Round 3 built it and measured 5.13-5.26 ms without, 5.26-5.56 ms with
# (the 11 saved reduce launches were paid back by the epilogues' 16 extra row reads per tile on a chain that was waiting for its weight gradients anyway);
# with the weight gradients on their own stream the data-gradient chain IS the critical path and the saved launches count: 3.70 -> 3.63 ms (same box, two
# alternations).  The sums are added in plan order instead of row order: gradients equal the separate pass to 1e-5 of each tensor's largest entry, and
# are reproducible run to run (one plan per table, fixed partial slots); 0 restores bit-identity with the per-module path.
BWD_SUMS_IN_CONV = os.environ.get("SEEVCN_BN_BWD_IN_CONV", "1") != "0"
# 0: every weight gradient is followed by its own slab-reduction launch (SV_OP_WGRAD) instead of ONE reduction launch for all layers at the end of the backward
# list (SV_OP_WGRAD_DEFERRED: bitwise the same gradients) -- A/B runs
DEFER_WGRAD_REDUCE = os.environ.get("SEEVCN_WGRAD_DEFER", "1") != "0"
# 1 (default since round 5): the weight gradients of the backward list on a stream of their own (sv_run_ops_two_streams), each behind the BatchNorm backward that
# makes its operand; the main chain (data gradient -> next BatchNorm backward -> ...) does not wait for them until the end of the list.  Their matrix-core work
# fills the bandwidth-bound BatchNorm launches and the prologues / tails of the data-gradient launches.  Round 4 measured the trained side alone 3.16 -> 3.06 ms
# but the pipelined step 4.07-4.19 -> 4.3-4.4 ms (a third stream on a GPU the one-stage input side kept busy) and left it off; with round 5's two-stage
# prefetch and shorter forward the same switch takes the step from 3.92 to 3.73 ms (same box, two alternations: profiles/r05_wstream_ab.txt).  0: one stream.
WGRAD_STREAM = os.environ.get("SEEVCN_WGRAD_STREAM", "1") != "0"
_wgrad_stream = {}
# 1 (default): the BatchNorm + ReLU behind a conv is NOT applied in a pass of its own -- the block keeps the raw conv output and the norm's (scale, shift), and the
# next block's convolution and weight gradient apply them as they gather the rows (sv_conv_next_input_norm): one read + one write of every activation
# tensor and one launch per block less; the values a consumer sees are bit for bit those of the separate pass.  Tensors that leave the chain (taps) are
# made when somebody reads them (LazyTap), the last block's in the list.  0: every block writes its normalised output (A/B runs, tests).
BN_FOLD = os.environ.get("SEEVCN_BN_FOLD", "1") != "0"
OP_CONV_PLANNED, OP_CONV_PLAIN, OP_BN_FWD, OP_BN_BWD, OP_WGRAD, OP_DGRAD_PLANNED_BN, OP_WGRAD_DEFERRED, OP_BN_FINALIZE, OP_BN_APPLY = 1, 2, 3, 5, 6, 7, 8, 9, 10
WORDS = 32


def _bits(x):
    return struct.unpack('<q', struct.pack('<d', float(x)))[0]


def _row(code, i=(), n=(), f=(), p=()):
    r = [0] * WORDS
    r[0] = code
    r[1:1 + len(i)] = [int(v) for v in i]
    r[9:9 + len(n)] = [int(v) for v in n]
    r[13:13 + len(f)] = f
    r[17:17 + len(p)] = [0 if v is None else int(v) for v in p]
    return r


def _run(rows, what):
    arr = np.array(rows, dtype=np.int64)
    _lib.check(_lib.load().sv_run_ops(arr.ctypes.data, len(rows), _lib.stream()), what)


def _run_two_streams(rows, dev):
    """The backward list with its weight gradients on a second stream (sv_run_ops_two_streams): the chain's data gradients and BatchNorm backwards do not
    wait for them; the current stream is ordered behind the side stream when the call returns."""
    side = _wgrad_stream.get(dev)
    if side is None:
        # SEEVCN_WGRAD_STREAM_PRIORITY (A/B): priority of the weight gradients' stream (torch: lower number = served first; default 0 = the main stream's)
        side = _wgrad_stream[dev] = torch.cuda.Stream(dev, priority=int(os.environ.get("SEEVCN_WGRAD_STREAM_PRIORITY", "0")))
    arr = np.array(rows, dtype=np.int64)
    _lib.check(_lib.load().sv_run_ops_two_streams(arr.ctypes.data, len(rows), _lib.stream(), side.cuda_stream), "sv_run_ops_two_streams (chain backward)")


class Block:
    """One conv -> norm (-> ReLU) block of a chain: the modules (parameters and running statistics stay theirs) and what is fixed about them."""

    def __init__(self, conv, bn, relu, tap):
        self.conv, self.bn, self.relu, self.tap = conv, bn, bool(relu), tap          # tap: this block's output is returned by the chain
        self.K = conv.kernel_size[0] * conv.kernel_size[1] * conv.kernel_size[2]
        self.cin, self.cout = conv.in_channels, conv.out_channels

    @property
    def mom_eps(self):
        # read when the launch list is made, i.e. after applicable() -> fusable_with() has rejected momentum=None (cumulative average)
        return (_bits(self.bn.momentum), _bits(self.bn.eps))


class BlockList(list):
    """The blocks of a chain + every module the flattening walked over (stage containers, nested SparseSequentials, convs, norms, ReLUs): the chain
    bypasses __call__ of ALL of them, so a hook on any of them must send the forward back to the module tree."""
    walked = ()


def _has_hooks(m):
    return bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, '_backward_pre_hooks', None))


def flatten_blocks(groups):
    """groups: the backbone's stages in execution order (SparseSequential each).  -> list of Block, or None when a stage is not a plain sequence of
    (SparseConvolution, BatchNorm1d, ReLU) triples.  The last block of every stage is a tap."""
    from .conv import SparseConvolution
    from .modules import SparseSequential
    blocks = BlockList()
    walked = []

    def walk(m, out):
        walked.append(m)
        for child in m._modules.values():
            if isinstance(child, SparseSequential):
                if not walk(child, out):
                    return False
            else:
                walked.append(child)
                out.append(child)
        return True

    for stage in groups:
        mods = []
        if not isinstance(stage, SparseSequential) or not walk(stage, mods) or len(mods) % 3 != 0 or not mods:
            return None
        for j in range(0, len(mods), 3):
            conv, bn, relu = mods[j:j + 3]
            if not (isinstance(conv, SparseConvolution) and type(bn) is torch.nn.BatchNorm1d and type(relu) is torch.nn.ReLU):
                return None
            blocks.append(Block(conv, bn, True, False))
        blocks[-1].tap = True
    blocks.walked = tuple(walked)
    return blocks


def applicable(blocks, x):
    """The chain takes these blocks on x now: training with gradients on, fp32 CUDA features, every block what fusable_with() accepts, no hooks,
    every rulebook in x's indice_dict (prebuild_rulebooks ran) with at least two output rows, and a planned data-gradient kernel for every block
    behind the first (the first one's is needed only when the input features want a gradient, which the backbone's never do)."""
    if CHAIN_OFF or blocks is None or not torch.is_grad_enabled() or x.features.requires_grad or x.indices.shape[0] < 2:
        return False
    if any(_has_hooks(m) for m in getattr(blocks, 'walked', ())):
        return False               # a forward / pre-forward / backward hook on a stage, a nested sequential, a conv, a norm or a ReLU: module path
    for k, b in enumerate(blocks):
        if not b.conv.fusable_with(b.bn, x) or b.conv.indice_key is None:
            return False
        rb = x.indice_dict.get(b.conv.indice_key)
        if rb is None or rb.n_out < 2 or rb.ksize != b.conv.kernel_size:
            return False
        if k > 0 and rb.plan_addrs("bwd", b.cout, b.cin) is None:
            return False
    return True


class _TapApply(torch.autograd.Function):
    """The normalised output of a chain block made on demand from its raw conv output: y = [relu](x * scale + shift) (sv_batchnorm_apply: the forward's
    own elementwise pass).  For autograd it is the identity: the chain's backward runs that block's BatchNorm backward itself, with whatever gradient
    arrives here as the gradient w.r.t. y."""

    @staticmethod
    def forward(ctx, raw, coef, relu):
        y = torch.empty_like(raw)
        _lib.check(_lib.load().sv_batchnorm_apply(_lib.ptr(raw), raw.shape[0], raw.shape[1], _lib.ptr(coef), int(relu), _lib.ptr(y), _lib.stream()), "sv_batchnorm_apply")
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def fold_plan(blocks, rulebooks):
    """-> (fold_in, materialize): fold_in[k]: block k reads block k-1's RAW conv output through that block's BatchNorm coefficients (its conv runs on a
    plan and its weight gradient on an MFMA tile shape: the kernels that carry the transform); materialize[k]: block k writes its normalised output in
    the list (the last block, and a block whose successor cannot fold)."""
    L = len(blocks)
    fold_in = [False] * L
    if BN_FOLD:
        for k in range(1, L):
            b, rb = blocks[k], rulebooks[k]
            fold_in[k] = rb.plan_addrs("fwd", b.cin, b.cout) is not None and b.cin % 16 == 0 and b.cout % 16 == 0
    materialize = [k == L - 1 or not fold_in[k + 1] for k in range(L)]
    return fold_in, materialize


class SparseChainFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, blocks, rulebooks, *params):
        lib = _lib.load()
        dev = features.device
        ctx.set_materialize_grads(False)                                              # a tap nobody differentiates arrives as None, not as zeros
        features = features.contiguous().float()
        fold_in, materialize = fold_plan(blocks, rulebooks)
        # one allocation for every activation: per block [conv output | block output (when it is written at all) | batch mean | batch invstd | scale | shift]
        offs, total = [], 0
        for k, (b, rb) in enumerate(zip(blocks, rulebooks)):
            n = rb.n_out * b.cout
            ny = n if materialize[k] else 0
            offs.append((total, total + n, total + n + ny, total + n + ny + b.cout, total + n + ny + 2 * b.cout))
            total += n + ny + 4 * b.cout
        arena = torch.empty((total,), dtype=torch.float32, device=dev)
        base = arena.data_ptr()
        n_part = lib.sv_conv_planned_partials()
        rows, x_ptr, n_src, keep = [], features.data_ptr(), features.shape[0], []
        frags = []
        in_coef = None                                                                # (address, relu) of the transform the next conv applies on load
        for k, (b, rb) in enumerate(zip(blocks, rulebooks)):
            w, gamma, beta = params[3 * k:3 * k + 3]
            o_conv, o_y, o_mean, o_istd, o_coef = (base + 4 * v for v in offs[k])
            wk = b.conv.weight_kio_nograd()
            plan = rb.plan_addrs("fwd", b.cin, b.cout)
            scratch = norm._scratch(b.cout, dev).data_ptr()
            partial = 0
            if plan is not None:
                a_rows, a_perm, a_masks_p, a_tiles, g, rev = plan
                ff, fb = Fsp.fragment_cache.get(wk)
                frags.append(fb)
                partial = scratch + 16 * b.cout if norm.STATS_IN_CONV else 0
                rows.append(_row(OP_CONV_PLANNED, i=(g, b.K, b.cin, b.cout, 0, int(bool(rev)), in_coef[1] if in_coef else 0), n=(n_src, rb.n_out),
                                 p=(x_ptr, a_rows, a_perm, a_masks_p, a_tiles, ff.data_ptr(), o_conv, None, None, None, None, partial or None,
                                    in_coef[0] if in_coef else None)))
            else:
                assert in_coef is None
                frags.append(None)
                wt = wk.detach().permute(0, 2, 1).contiguous()                       # (K, C_out, C_in): the 3-channel input layer only
                keep.append(wt)
                rows.append(_row(OP_CONV_PLAIN, i=(b.K, b.cin, b.cout, 0), n=(n_src, rb.n_out), p=(x_ptr, rb.addr("nbr_out"), wt.data_ptr(), o_conv)))
            if BN_FOLD:
                rows.append(_row(OP_BN_FINALIZE, i=(b.cout, n_part if partial else 0), n=(rb.n_out,), f=b.mom_eps,
                                 p=(gamma.data_ptr(), beta.data_ptr(), b.bn.running_mean.data_ptr(), b.bn.running_var.data_ptr(), scratch, o_coef, o_mean, o_istd,
                                    b.bn.num_batches_tracked.data_ptr(), None if partial else o_conv)))
                if materialize[k]:
                    rows.append(_row(OP_BN_APPLY, i=(b.cout, int(b.relu)), n=(rb.n_out,), p=(o_conv, o_coef, o_y)))
            else:
                rows.append(_row(OP_BN_FWD, i=(b.cout, 1, int(b.relu), n_part if partial else 0), n=(rb.n_out,), f=b.mom_eps,
                                 p=(o_conv, gamma.data_ptr(), beta.data_ptr(), b.bn.running_mean.data_ptr(), b.bn.running_var.data_ptr(), scratch, o_y, o_mean,
                                    o_istd, b.bn.num_batches_tracked.data_ptr())))
            if k + 1 < len(blocks) and fold_in[k + 1]:
                x_ptr, in_coef = o_conv, (o_coef, int(b.relu))
            else:
                x_ptr, in_coef = o_y, None
            n_src = rb.n_out
        _run(rows, "sv_run_ops (chain forward)")
        ctx.blocks, ctx.rulebooks, ctx.offs, ctx.frags, ctx.fold_in = blocks, rulebooks, offs, frags, fold_in
        ctx.save_for_backward(features, arena, *params)
        # a tap whose normalised output is written in the list is that tensor; the others hand out [raw conv output, coefficients] (see run_chain)
        outs = []
        for k, b in enumerate(blocks):
            if b.tap:
                n, c = rulebooks[k].n_out, b.cout
                if materialize[k]:
                    outs.append(arena[offs[k][1]:offs[k][2]].view(n, c))
                else:
                    outs.append(arena[offs[k][0]:offs[k][1]].view(n, c))
        coefs = [arena[offs[k][4]:offs[k][4] + 2 * b.cout] for k, b in enumerate(blocks) if b.tap and not materialize[k]]
        ctx.mark_non_differentiable(*coefs)
        return tuple(outs) + tuple(coefs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        features, arena, *params = ctx.saved_tensors
        blocks, rulebooks, offs, frags, fold_in = ctx.blocks, ctx.rulebooks, ctx.offs, ctx.frags, ctx.fold_in
        dev = arena.device
        L = len(blocks)
        ext, gi = [None] * L, 0                                                       # gradient that reaches a block's output from outside the chain
        for k, b in enumerate(blocks):                                                # (behind the taps' gradients come the Nones of the coefficient outputs)
            if b.tap:
                ext[k] = None if grads[gi] is None else grads[gi].contiguous().float()
                gi += 1
        if ext[L - 1] is None:
            ext[L - 1] = torch.zeros((rulebooks[-1].n_out, blocks[-1].cout), dtype=torch.float32, device=dev)
        # one allocation for the work buffers: per block [gradient of the conv output | gradient of the block's input (blocks >= 1) | dgamma | dbeta],
        # one for the weight gradients (in the parameters' own layout)
        boffs, total, woffs, wtotal, wbytes, poffs, wplans = [], 0, [], 0, 0, [], []
        for k, (b, rb) in enumerate(zip(blocks, rulebooks)):
            n, nin = rb.n_out * b.cout, (rb.n_in * b.cin if k > 0 else 0)
            boffs.append((total, total + n, total + n + nin, total + n + nin + b.cout))
            total += n + nin + 2 * b.cout
            woffs.append(wtotal)
            wtotal += b.K * b.cin * b.cout
            wp = rb.wgrad_plan(b.cin, b.cout)                        # equal-pieces plan of the table (built with the index; here only if it was not)
            wplans.append(0 if wp is None else wp.data_ptr())
            if DEFER_WGRAD_REDUCE:                                   # every layer keeps its own partial slabs until the one reduction at the end of the list
                poffs.append(wbytes)
                wbytes += lib.sv_sparse_conv_wgrad_partial_bytes(rb.n_out, b.K, b.cin, b.cout) if wp is None else lib.sv_sparse_conv_wgrad_planned_bytes(b.K, b.cin, b.cout)
            else:
                poffs.append(0)
                wbytes = max(wbytes, lib.sv_sparse_conv_wgrad_scratch_bytes(rb.n_out, b.K, b.cin, b.cout) if wp is None
                             else lib.sv_sparse_conv_wgrad_planned_bytes(b.K, b.cin, b.cout))
        work = torch.empty((total,), dtype=torch.float32, device=dev)
        wgrads = torch.empty((wtotal,), dtype=torch.float32, device=dev)
        wscratch = _lib.workspace.scratch("wgrad_layers" if DEFER_WGRAD_REDUCE else "wgrad", wbytes, dev)
        base, abase, wbase = work.data_ptr(), arena.data_ptr(), wgrads.data_ptr()
        rows = []
        n_part, n_part_bwd = lib.sv_conv_planned_partials(), 0          # n_part_bwd: partials the data-gradient launch above left for this BatchNorm
        dy_ptr = ext[L - 1].data_ptr()
        for k in range(L - 1, -1, -1):
            b, rb = blocks[k], rulebooks[k]
            w, gamma, beta = params[3 * k:3 * k + 3]
            o_dconv, o_dx, o_dg, o_db = (base + 4 * v for v in boffs[k])
            a_conv, a_y, a_mean, a_istd, _ = (abase + 4 * v for v in offs[k])
            # the layer's input: the features, the block below's normalised output, or (folded) its raw conv output + the coefficients of its BatchNorm
            x_in = features.data_ptr() if k == 0 else abase + 4 * offs[k - 1][0 if fold_in[k] else 1]
            x_coef = (abase + 4 * offs[k - 1][4], int(blocks[k - 1].relu)) if fold_in[k] else (None, 0)
            scratch = norm._scratch(b.cout, dev)
            rows.append(_row(OP_BN_BWD, i=(b.cout, int(b.relu), n_part_bwd), n=(rb.n_out,),
                             p=(a_conv, dy_ptr, gamma.data_ptr(), beta.data_ptr(), a_mean, a_istd, scratch.data_ptr(), o_dconv, o_dg, o_db)))
            n_part_bwd = 0
            rows.append(_row(OP_WGRAD_DEFERRED if DEFER_WGRAD_REDUCE else OP_WGRAD, i=(b.K, b.cin, b.cout, rb.n_in, x_coef[1]), n=(rb.n_out, b.cin, 1, b.K * b.cin),
                             p=(x_in, rb.addr("nbr_out"), o_dconv, wbase + 4 * woffs[k], wscratch.data_ptr() + poffs[k], wplans[k] or None, x_coef[0])))
            if k > 0:
                a_rows, a_perm, a_masks_p, a_tiles, g, rev = rb.plan_addrs("bwd", b.cout, b.cin)
                res = ext[k - 1]
                if res is None and norm.STATS_IN_CONV and BWD_SUMS_IN_CONV:
                    # the gradient this launch writes is the whole gradient of block k-1's output: its epilogue also makes the two sums of that
                    # block's BatchNorm backward (the rows' x comes from the arena), and the BatchNorm op below starts at the combine
                    lo = blocks[k - 1]
                    p_conv, _, p_mean, p_istd, _ = (abase + 4 * v for v in offs[k - 1])
                    g_lo, b_lo = params[3 * (k - 1) + 1], params[3 * (k - 1) + 2]
                    partial = norm._scratch(lo.cout, dev).data_ptr() + 16 * lo.cout
                    rows.append(_row(OP_DGRAD_PLANNED_BN, i=(g, b.K, b.cout, b.cin, int(bool(rev)), int(lo.relu)), n=(rb.n_out, rb.n_in),
                                     p=(o_dconv, a_rows, a_perm, a_masks_p, a_tiles, frags[k].data_ptr(), o_dx,
                                        p_conv, p_mean, p_istd, g_lo.data_ptr(), b_lo.data_ptr(), partial)))
                    n_part_bwd = n_part
                else:
                    rows.append(_row(OP_CONV_PLANNED, i=(g, b.K, b.cout, b.cin, 0, int(bool(rev))), n=(rb.n_out, rb.n_in),
                                     p=(o_dconv, a_rows, a_perm, a_masks_p, a_tiles, frags[k].data_ptr(), o_dx, None,
                                        None, None, None if res is None else res.data_ptr(), None)))
                dy_ptr = o_dx
        if WGRAD_STREAM:
            _run_two_streams(rows, dev)
        else:
            _run(rows, "sv_run_ops (chain backward)")
        out = [None, None, None]
        for k, b in enumerate(blocks):
            w = params[3 * k]
            o = boffs[k]
            out += [wgrads[woffs[k]:woffs[k] + b.K * b.cin * b.cout].view(w.shape), work[o[2]:o[3]], work[o[3]:o[3] + b.cout]]
        return tuple(out)


class LazyTap(SparseConvTensor):
    """A tap whose normalised features are made when somebody reads them (BN_FOLD): PV-RCNN's set abstraction reads multi_scale_3d_features, SECOND /
    the benchmarked step read none of x_conv1..4 -- their elementwise passes never run.  (One class for all taps: a class made per call is a
    reference cycle that only the cyclic collector frees, and it kept the tap's tensors alive with it -- sporadic allocator growth in the step.)"""

    def __init__(self, raw, coef, relu, indices, spatial_shape, batch_size, grid=None, indice_dict=None):
        super().__init__(None, indices, spatial_shape, batch_size, grid, indice_dict)
        self._raw, self._coef, self._relu = raw, coef, relu

    @property
    def features(self):
        if self._features is None:
            self._features = _TapApply.apply(self._raw, self._coef, self._relu)
            self._raw = self._coef = None
        return self._features

    @features.setter
    def features(self, value):
        self._features = value


def run_chain(blocks, x):
    """x: SparseConvTensor at the chain's input with every rulebook prebuilt.  -> list of SparseConvTensor, one per tap, in order."""
    rulebooks = [x.indice_dict[b.conv.indice_key] for b in blocks]
    params = []
    for b in blocks:
        params += [b.conv.weight, b.bn.weight, b.bn.bias]
    outs = SparseChainFunction.apply(x.features, blocks, rulebooks, *params)
    taps = [(b, rb) for b, rb in zip(blocks, rulebooks) if b.tap]
    _, materialize = fold_plan(blocks, rulebooks)
    tap_mat = [materialize[k] for k, b in enumerate(blocks) if b.tap]
    coefs = list(outs[len(taps):])
    res = []
    for f, (b, rb), mat in zip(outs[:len(taps)], taps, tap_mat):
        if mat:
            res.append(SparseConvTensor(f, rb.out_indices, rb.out_shape, x.batch_size, x.grid, x.indice_dict))
        else:
            res.append(LazyTap(f, coefs.pop(0), b.relu, rb.out_indices, rb.out_shape, x.batch_size, x.grid, x.indice_dict))
    return res
