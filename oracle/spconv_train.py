"""Oracle: the TRAINING step of VoxelBackBone8x + HeightCompression as one differentiable chain on the CPU.  Test infrastructure only.

The sparse convolutions are oracle/spconv.py's numpy restatement (spconv itself is un-vendored: PARITY UNPINNED, see that header) wrapped in a
torch.autograd.Function; BatchNorm1d(eps 1e-3, momentum 0.01, batch statistics) and ReLU are torch's own CPU ops in the chosen dtype -- the
reference's layer composition, detector3d/pcdet/models/backbones_3d/spconv_backbone.py:8-27,77-117,128-180; the dense scatter follows
backbones_2d/map_to_bev/height_compression.py:21-26.  float64 gives the tight reference for the gradient-parity test
(tests/test_spconv.py), float32 is what bench.py's cpu_baseline times."""
import numpy as np
import torch

from . import spconv as osp


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, nbr):
        ctx.nbr = nbr
        ctx.save_for_backward(x, w)
        return torch.from_numpy(osp.conv_forward(x.numpy(), nbr, w.numpy())).to(x.dtype)

    @staticmethod
    def backward(ctx, go):
        x, w = ctx.saved_tensors
        gf, gw = osp.conv_backward(x.numpy(), ctx.nbr, w.numpy(), go.numpy())
        return torch.from_numpy(gf).to(x.dtype), torch.from_numpy(gw).to(w.dtype), None


def backbone8x_train_chain(sd, features, coords, batch_size, sparse_shape, dtype=torch.float64):
    """sd: VoxelBackBone8x state_dict (numpy, 2.x weight layout, the reference's key names).  Returns (dense (B, C*D, H, W) tensor with grad_fn,
    leaves): leaves = {'input': features leaf, '<conv key>': (K, C_in, C_out) weight leaf, '<bn key>.weight' / '.bias': leaves}."""
    leaves = {}

    def W(key):
        t = torch.from_numpy(osp.weight_to_kio(np.asarray(sd[key]))).to(dtype).requires_grad_(True)
        leaves[key] = t
        return t

    def bn_relu(x, prefix):
        g = torch.from_numpy(np.asarray(sd[prefix + ".weight"])).to(dtype).requires_grad_(True)
        b = torch.from_numpy(np.asarray(sd[prefix + ".bias"])).to(dtype).requires_grad_(True)
        leaves[prefix + ".weight"], leaves[prefix + ".bias"] = g, b
        return torch.relu(torch.nn.functional.batch_norm(x, None, None, g, b, True, 0.01, 1e-3))

    shape = tuple(int(s) for s in sparse_shape)
    x = torch.from_numpy(np.asarray(features)).to(dtype).requires_grad_(True)
    leaves["input"] = x
    nb = osp.rulebook_subm(coords, shape, 3)
    x = bn_relu(_Conv.apply(x, W("conv_input.0.weight"), nb), "conv_input.1")
    x = bn_relu(_Conv.apply(x, W("conv1.0.0.weight"), nb), "conv1.0.1")
    c = coords
    for name, pad in (("conv2", 1), ("conv3", 1), ("conv4", (0, 1, 1))):
        oc, nbo, _, oshape = osp.rulebook_sparse(c, shape, 3, 2, pad)
        x = bn_relu(_Conv.apply(x, W(f"{name}.0.0.weight"), nbo), f"{name}.0.1")
        c, shape = oc, oshape
        nb = osp.rulebook_subm(c, shape, 3)
        for i in (1, 2):
            x = bn_relu(_Conv.apply(x, W(f"{name}.{i}.0.weight"), nb), f"{name}.{i}.1")
    oc, nbo, _, oshape = osp.rulebook_sparse(c, shape, (3, 1, 1), (2, 1, 1), 0)
    x = bn_relu(_Conv.apply(x, W("conv_out.0.weight"), nbo), "conv_out.1")
    dense = torch.zeros(batch_size, *oshape, x.shape[1], dtype=dtype)
    idx = [torch.from_numpy(oc[:, i].astype(np.int64)) for i in range(4)]
    dense = dense.index_put((idx[0], idx[1], idx[2], idx[3]), x)                       # (B, D, H, W, C)
    dense = dense.permute(0, 4, 1, 2, 3).reshape(batch_size, x.shape[1] * oshape[0], oshape[1], oshape[2])
    return dense, leaves, (x, oc, oshape)


def stage_train_chain(sd, layers, features, coords, shape, dtype=torch.float64, branch_hints=None, hint_band=0.0):
    """One STAGE of the backbone (a run of conv -> BatchNorm(batch statistics) -> ReLU blocks, spconv_backbone.py:8-27) as a differentiable chain.
    layers: [(conv key, bn prefix, 'subm' | 'sparse', ksize, stride, padding)], sd as in backbone8x_train_chain.  Returns (output tensor with grad_fn,
    leaves incl. 'input', out_coords, out_shape, n_overridden) -- used by the 16-scene sampled-stage gradient test, where the inputs and the
    upstream gradient of the stage are the GPU step's own tensors.

    The ReLU's derivative jumps at zero: for a pre-activation within rounding distance of zero a float32 and a float64 evaluation may take
    different branches, both valid, and the gradients of everything it feeds then differ by O(1).  branch_hints {conv key: bool (rows, C)} are
    the branches the evaluation under test took (its ReLU output > 0); they are followed ONLY where |pre-activation| < hint_band, everywhere
    else the oracle's own sign decides.  n_overridden counts the positions where the hint changed the oracle's branch."""
    leaves = {}
    x = torch.from_numpy(np.asarray(features)).to(dtype).requires_grad_(True)
    leaves["input"] = x
    c, shape = np.asarray(coords), tuple(int(s) for s in shape)
    overridden = 0
    for key, bn, kind, ksize, stride, pad in layers:
        w = torch.from_numpy(osp.weight_to_kio(np.asarray(sd[key]))).to(dtype).requires_grad_(True)
        g = torch.from_numpy(np.asarray(sd[bn + ".weight"])).to(dtype).requires_grad_(True)
        b = torch.from_numpy(np.asarray(sd[bn + ".bias"])).to(dtype).requires_grad_(True)
        leaves[key], leaves[bn + ".weight"], leaves[bn + ".bias"] = w, g, b
        if kind == "subm":
            nbr = osp.rulebook_subm(c, shape, ksize)
        else:
            c, nbr, _, shape = osp.rulebook_sparse(c, shape, ksize, stride, pad)
        z = torch.nn.functional.batch_norm(_Conv.apply(x, w, nbr), None, None, g, b, True, 0.01, 1e-3)
        on = z.detach() > 0
        if branch_hints is not None and key in branch_hints:
            hint = torch.from_numpy(np.asarray(branch_hints[key], bool))
            use = (z.detach().abs() < hint_band) & (hint != on)
            overridden += int(use.sum())
            on = torch.where(use, hint, on)
        x = torch.where(on, z, torch.zeros_like(z))
    return x, leaves, c, shape, overridden
