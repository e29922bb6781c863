"""Oracle: sparse 3-D convolution semantics (numpy).  Test infrastructure only.

PARITY UNPINNED: the reference takes these operations from the third-party package `spconv`
(pip `spconv-cu102`, no version pin: docker/Dockerfile:58; 1.x and 2.x both tolerated,
detector3d/pcdet/utils/spconv_utils.py:3-6), whose source is not under /root/reference and which is not
installed here; the reference has no test or golden vector at this boundary.
This file restates spconv's published semantics:

  SubMConv3d   out[i] = sum_k W[k] . in[j]  where coord[j] = coord[i] + (k - K//2)*dilation; output rows = input rows.
  SparseConv3d out coordinate o = (i + pad - k*dilation)/stride where divisible and 0 <= o < out_shape,
               out_shape = floor((D + 2*pad - dilation*(K-1) - 1)/stride) + 1; output set = unique of those.
  Output order (implementation-defined in spconv) is fixed here to ascending key ((b*Z+z)*Y+y)*X+x.
  Weights: W[kz,ky,kx,c_in,c_out] (spconv 1.x layout) <-> (c_out,kz,ky,kx,c_in) (2.x),
           detector3d/pcdet/models/detectors/detector3d_template.py:341-348.

Call sites anchored: detector3d/pcdet/models/backbones_3d/spconv_backbone.py:77-117 (layers), :141-157 (forward),
backbones_2d/map_to_bev/height_compression.py:21-23 (dense).
"""
import numpy as np


def _triple(v):
    return tuple(int(x) for x in (v if np.ndim(v) else (v, v, v)))


def out_shape(in_shape, ksize, stride, padding, dilation=1):
    k, s, p, d = _triple(ksize), _triple(stride), _triple(padding), _triple(dilation)
    return tuple((int(in_shape[i]) + 2 * p[i] - d[i] * (k[i] - 1) - 1) // s[i] + 1 for i in range(3))


def _key(b, z, y, x, shape):
    return ((b.astype(np.int64) * shape[0] + z) * shape[1] + y) * shape[2] + x


def _offsets(ksize):
    kz, ky, kx = _triple(ksize)
    return [(a, b, c) for a in range(kz) for b in range(ky) for c in range(kx)]


def rulebook_subm(coords, shape, ksize, dilation=1):
    """nbr (K, N) int32: input row feeding output row i through offset k, or -1."""
    coords = np.asarray(coords, np.int64)
    n = len(coords)
    k3, d3 = _triple(ksize), _triple(dilation)
    keys = _key(coords[:, 0], coords[:, 1], coords[:, 2], coords[:, 3], shape)
    order = np.argsort(keys, kind="stable")
    skeys = keys[order]
    nbr = np.full((len(_offsets(ksize)), n), -1, np.int32)
    for k, (a, b, c) in enumerate(_offsets(ksize)):
        z = coords[:, 1] + (a - k3[0] // 2) * d3[0]
        y = coords[:, 2] + (b - k3[1] // 2) * d3[1]
        x = coords[:, 3] + (c - k3[2] // 2) * d3[2]
        ok = (z >= 0) & (z < shape[0]) & (y >= 0) & (y < shape[1]) & (x >= 0) & (x < shape[2])
        q = _key(coords[:, 0], z, y, x, shape)
        pos = np.searchsorted(skeys, q)
        pos = np.minimum(pos, max(n - 1, 0))
        hit = ok & (n > 0) & (skeys[pos] == q)
        nbr[k, hit] = order[pos[hit]].astype(np.int32)
    return nbr


def rulebook_sparse(coords, in_shape, ksize, stride, padding, dilation=1):
    """Returns (out_coords (M,4) int32 canonical order, nbr_out (K,M) int32, nbr_in (K,N) int32, out_shape)."""
    coords = np.asarray(coords, np.int64)
    n = len(coords)
    s3, p3, d3 = _triple(stride), _triple(padding), _triple(dilation)
    oshape = out_shape(in_shape, ksize, stride, padding, dilation)
    offs = _offsets(ksize)
    cand_key = np.full((len(offs), n), -1, np.int64)
    for k, (a, b, c) in enumerate(offs):
        tz = coords[:, 1] + p3[0] - a * d3[0]
        ty = coords[:, 2] + p3[1] - b * d3[1]
        tx = coords[:, 3] + p3[2] - c * d3[2]
        ok = (tz >= 0) & (ty >= 0) & (tx >= 0) & (tz % s3[0] == 0) & (ty % s3[1] == 0) & (tx % s3[2] == 0)
        oz, oy, ox = tz // s3[0], ty // s3[1], tx // s3[2]
        ok &= (oz < oshape[0]) & (oy < oshape[1]) & (ox < oshape[2])
        cand_key[k, ok] = _key(coords[:, 0], oz, oy, ox, oshape)[ok]
    uniq = np.unique(cand_key[cand_key >= 0])
    m = len(uniq)
    nbr_in = np.full((len(offs), n), -1, np.int32)
    valid = cand_key >= 0
    nbr_in[valid] = np.searchsorted(uniq, cand_key[valid]).astype(np.int32)
    nbr_out = np.full((len(offs), m), -1, np.int32)
    for k in range(len(offs)):
        v = nbr_in[k] >= 0
        nbr_out[k, nbr_in[k, v]] = np.nonzero(v)[0].astype(np.int32)
    x = uniq % oshape[2]
    t = uniq // oshape[2]
    y = t % oshape[1]
    t //= oshape[1]
    z = t % oshape[0]
    b = t // oshape[0]
    out_coords = np.stack([b, z, y, x], axis=1).astype(np.int32)
    return out_coords, nbr_out, nbr_in, oshape


def conv_forward(features, nbr_out, weight, bias=None):
    """features (N_in, C_in), weight (K, C_in, C_out) -> (N_out, C_out); fp64 accumulation for a tight reference."""
    k, m = nbr_out.shape
    out = np.zeros((m, weight.shape[2]), np.float64)
    f = np.asarray(features, np.float64)
    for kk in range(k):
        v = nbr_out[kk] >= 0
        if v.any():
            out[v] += f[nbr_out[kk, v]] @ np.asarray(weight[kk], np.float64)
    if bias is not None:
        out += np.asarray(bias, np.float64)
    return out


def conv_backward(features, nbr_out, weight, grad_out):
    """Returns (grad_features (N_in,C_in), grad_weight (K,C_in,C_out)) in fp64."""
    k, m = nbr_out.shape
    f = np.asarray(features, np.float64)
    g = np.asarray(grad_out, np.float64)
    gf = np.zeros_like(f)
    gw = np.zeros(weight.shape, np.float64)
    for kk in range(k):
        v = nbr_out[kk] >= 0
        if v.any():
            j = nbr_out[kk, v]
            np.add.at(gf, j, g[v] @ np.asarray(weight[kk], np.float64).T)
            gw[kk] = f[j].T @ g[v]
    return gf, gw


def dense(features, coords, batch_size, spatial_shape):
    """SparseConvTensor.dense(): (B, C, D, H, W)."""
    c = features.shape[1]
    out = np.zeros((batch_size, c) + tuple(int(s) for s in spatial_shape), features.dtype)
    out[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]] = features
    return out


def pair_counts(nbr_out):
    return (nbr_out >= 0).sum(axis=1).astype(np.int32)


def weight_to_kio(weight_2x):
    """(C_out,kz,ky,kx,C_in) -> (K, C_in, C_out)."""
    co, kz, ky, kx, ci = weight_2x.shape
    return np.ascontiguousarray(np.transpose(weight_2x.reshape(co, kz * ky * kx, ci), (1, 2, 0)))


def _bn_relu(x, sd, prefix, eps=1e-3):
    """Eval-mode BatchNorm1d(eps=1e-3, spconv_backbone.py:73) + ReLU on (N,C)."""
    g, b, m, v = (np.asarray(sd[f"{prefix}.{k}"], np.float64) for k in ("weight", "bias", "running_mean", "running_var"))
    return np.maximum((x - m) / np.sqrt(v + eps) * g + b, 0.0)


def voxel_backbone8x_forward(sd, features, coords, batch_size, sparse_shape):
    """VoxelBackBone8x.forward (spconv_backbone.py:128-180) in eval mode; returns dict of (features, coords, shape)
    for x_conv1..4 and out.  sd holds the 2.x-layout weights under the reference's key names."""
    def subm(x, c, shape, key, cache, ck):
        if ck not in cache:
            cache[ck] = rulebook_subm(c, shape, 3)
        return conv_forward(x, cache[ck], weight_to_kio(np.asarray(sd[key + ".0.weight"])))

    cache = {}
    x = _bn_relu(subm(features, coords, sparse_shape, "conv_input", cache, "subm1"), sd, "conv_input.1")
    x = _bn_relu(subm(x, coords, sparse_shape, "conv1.0", cache, "subm1"), sd, "conv1.0.1")
    res = {"x_conv1": (x, coords, tuple(sparse_shape))}
    c, shape = coords, tuple(sparse_shape)
    for name, pad in (("conv2", 1), ("conv3", 1), ("conv4", (0, 1, 1))):
        oc, nbr_out, _, oshape = rulebook_sparse(c, shape, 3, 2, pad)
        x = _bn_relu(conv_forward(x, nbr_out, weight_to_kio(np.asarray(sd[f"{name}.0.0.weight"]))), sd, f"{name}.0.1")
        c, shape = oc, oshape
        for i in (1, 2):
            x = _bn_relu(subm(x, c, shape, f"{name}.{i}", cache, name), sd, f"{name}.{i}.1")
        res["x_" + name] = (x, c, shape)
    oc, nbr_out, _, oshape = rulebook_sparse(c, shape, (3, 1, 1), (2, 1, 1), 0)
    x = _bn_relu(conv_forward(x, nbr_out, weight_to_kio(np.asarray(sd["conv_out.0.weight"]))), sd, "conv_out.1")
    res["out"] = (x, oc, oshape)
    return res


def voxel_res_backbone8x_forward(sd, features, coords, batch_size, sparse_shape):
    """VoxelResBackBone8x.forward (spconv_backbone.py:241-293, layers :191-232; SparseBasicBlock :30-66) in eval mode: conv_input,
    two residual blocks per level (conv(bias) - bn - relu - conv(bias) - bn, + identity, relu; both convs of a block and both blocks of a
    level share one submanifold rulebook), strided conv-bn-relu between levels, conv_out (3,1,1)/(2,1,1).  Returns {name: (features, coords,
    shape)} for x_conv1..4 and out; sd holds 2.x-layout weights under the reference's key names."""
    def w(key):
        return weight_to_kio(np.asarray(sd[key]))

    def basic_block(x, nbr, prefix):
        y = conv_forward(x, nbr, w(prefix + ".conv1.weight"), np.asarray(sd[prefix + ".conv1.bias"], np.float64))
        y = _bn_relu(y, sd, prefix + ".bn1")
        y = conv_forward(y, nbr, w(prefix + ".conv2.weight"), np.asarray(sd[prefix + ".conv2.bias"], np.float64))
        g, b, m, v = (np.asarray(sd[f"{prefix}.bn2.{k}"], np.float64) for k in ("weight", "bias", "running_mean", "running_var"))
        y = (y - m) / np.sqrt(v + 1e-3) * g + b
        return np.maximum(y + x, 0.0)

    shape = tuple(int(s) for s in sparse_shape)
    nbr = rulebook_subm(coords, shape, 3)
    x = _bn_relu(conv_forward(features, nbr, w("conv_input.0.weight")), sd, "conv_input.1")
    for i in (0, 1):
        x = basic_block(x, nbr, f"conv1.{i}")
    res = {"x_conv1": (x, coords, shape)}
    c = coords
    for name, pad in (("conv2", 1), ("conv3", 1), ("conv4", (0, 1, 1))):
        oc, nbr_out, _, oshape = rulebook_sparse(c, shape, 3, 2, pad)
        x = _bn_relu(conv_forward(x, nbr_out, w(f"{name}.0.0.weight")), sd, f"{name}.0.1")
        c, shape = oc, oshape
        nbr = rulebook_subm(c, shape, 3)
        for i in (1, 2):
            x = basic_block(x, nbr, f"{name}.{i}")
        res["x_" + name] = (x, c, shape)
    oc, nbr_out, _, oshape = rulebook_sparse(c, shape, (3, 1, 1), (2, 1, 1), 0)
    x = _bn_relu(conv_forward(x, nbr_out, w("conv_out.0.weight")), sd, "conv_out.1")
    res["out"] = (x, oc, oshape)
    return res
