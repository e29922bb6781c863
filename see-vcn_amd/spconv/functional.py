"""Host wrappers + autograd Functions over the rulebook / sparse-conv entry points of libseevcn_hip.so."""
import ctypes
import os

import torch

from .. import _lib


def _i3(v):
    v = list(v) if isinstance(v, (list, tuple)) else [v, v, v]
    assert len(v) == 3
    return _lib.host_array(ctypes.c_int32, [int(x) for x in v])


def conv_out_shape(in_shape, ksize, stride, padding, dilation):
    lib = _lib.load()
    out = (ctypes.c_int32 * 3)()
    _lib.check(lib.sv_conv_out_shape(_i3(in_shape), _i3(ksize), _i3(stride), _i3(padding), _i3(dilation), out), "sv_conv_out_shape")
    return [int(x) for x in out]


class Rulebook:
    """Output-major table nbr_out (K, N_out) plus, for strided convs, the input-major nbr_in (K, N_in)."""

    def __init__(self, nbr_out, nbr_in, out_indices, out_shape, n_in, n_out, subm, ksize):
        self.nbr_out, self.nbr_in = nbr_out, nbr_in
        self.out_indices, self.out_shape = out_indices, out_shape
        self.n_in, self.n_out, self.subm, self.ksize = n_in, n_out, subm, ksize
        self._nbr_in_subm = None
        self._orders = {}
        self._groups = {}
        self._inverse = None
        self.in_indices = self.in_shape = None      # set by the conv that built the rulebook

    @property
    def K(self):
        return self.nbr_out.shape[0]

    def inverse_view(self):
        """The same pairs read the other way round (SparseInverseConv3d): the input-major table becomes the output-major one."""
        assert not self.subm and self.nbr_in is not None
        if self._inverse is None:
            self._inverse = Rulebook(self.nbr_in, self.nbr_out, self.in_indices, self.in_shape, self.n_out, self.n_in, False, self.ksize)
        return self._inverse

    def table_for_backward_data(self):
        """Input-major table. For SubM it is the output-major table with the offsets reversed
        (coord[j] = coord[i] + d  <=>  coord[i] = coord[j] - d)."""
        if not self.subm:
            return self.nbr_in
        if self._nbr_in_subm is None:
            self._nbr_in_subm = torch.flip(self.nbr_out, dims=[0]).contiguous()
        return self._nbr_in_subm

    def tile_order(self, table, kd, nc):
        """Work-balanced tile order of one of this rulebook's tables (sv_conv_tile_order) for a (kd -> nc)-channel gather-GEMM,
        computed once and reused by every launch with the same tiles-per-wave on that table."""
        lib = _lib.load()
        n_rows, K = table.shape[1], table.shape[0]
        if not TILE_ORDER or n_rows == 0 or (int(kd) // 16) * (int(nc) // 16) < 4 or kd % 16 or nc % 16:
            return None              # load-bound small-channel layers: the order does not pay for its own launch
        g = lib.sv_conv_tiles_per_wave(n_rows, int(kd), int(nc))
        if self.subm and self._nbr_in_subm is not None and table.data_ptr() == self._nbr_in_subm.data_ptr():
            table = self.nbr_out     # the flipped table has the same number of active offsets per tile: one order serves both directions
        key = (table.data_ptr(), g)
        if key not in self._orders:
            order = torch.empty((lib.sv_conv_tile_order_bytes(n_rows) // 4,), dtype=torch.int32, device=table.device)
            scratch = _lib.workspace.scratch("tile_order", lib.sv_conv_tile_order_scratch_bytes(n_rows), table.device)
            _lib.check(lib.sv_conv_tile_order(_lib.ptr(table), n_rows, K, g, _lib.ptr(scratch), _lib.ptr(order), _lib.stream()), "sv_conv_tile_order")
            self._orders[key] = (order, table)       # keep the table alive with its order
        return self._orders[key][0]

    def plan(self, direction, kd, nc):
        """(table, tile_order, row_perm, table_k_reversed) for a (kd -> nc)-channel gather-GEMM over this rulebook; direction 'fwd' (output-major
        table) or 'bwd' (input-major table).  When the MFMA kernel applies (channels multiples of 16 up to 64, K <= 27) the
        table's columns are regrouped by neighbour mask: row_perm[p] is the row that column p of the returned table produces.
        A 16-row tile executes an offset when ANY of its rows has that neighbour; tiles of consecutive rows waste 40-80 % of
        those steps (70 % on a strided conv's input-major table, whose masks are a function of coordinate parity), tiles of
        equal-mask rows almost none.  Computed once per table and reused by every launch on it."""
        assert direction in ("fwd", "bwd")
        kd, nc = int(kd), int(nc)
        n_rows = self.n_out if direction == "fwd" else self.n_in
        mfma_kernel = bool(_lib.load().sv_conv_mfma_kernel_applies(int(self.K), kd, nc))              # k_spconv_rs3 takes the layer
        # narrow layers are bound by their loads, and reading the table through the permutation costs them more than equal-mask tiles save
        groupable = GROUP_ROWS and mfma_kernel and max(kd, nc) >= 64 and 64 <= n_rows <= 16 * 65536
        if not groupable:
            if direction == "bwd" and self.subm and mfma_kernel:
                # the submanifold table read with its offsets reversed IS its input-major table: no flipped copy, same tile order
                return self.nbr_out, self.tile_order(self.nbr_out, kd, nc), None, True
            table = self.nbr_out if direction == "fwd" else self.table_for_backward_data()
            return table, self.tile_order(table, kd, nc), None, False
        key = "fwd" if (direction == "fwd" or self.subm) else "bwd"    # a submanifold table serves its own data gradient reversed
        lib = _lib.load()
        base = self.nbr_out if key == "fwd" else self.nbr_in
        if key not in self._groups:
            perm = torch.empty((n_rows,), dtype=torch.int32, device=base.device)
            masks = torch.empty((n_rows,), dtype=torch.int32, device=base.device)
            hist = _lib.workspace.persistent("conv_group_hist", lib.sv_conv_group_persistent_bytes(), base.device)
            _lib.check(lib.sv_conv_group_rows(_lib.ptr(base), n_rows, self.K, _lib.ptr(hist), _lib.ptr(masks), _lib.ptr(perm), _lib.stream()),
                       "sv_conv_group_rows")
            self._groups[key] = (perm, masks)
        perm, masks = self._groups[key]
        if (kd // 16) * (nc // 16) < 4:
            return base, None, perm, (direction == "bwd" and self.subm)       # load-bound layers: equal-mask tiles already cost the same
        g = lib.sv_conv_tiles_per_wave(n_rows, kd, nc)
        okey = ("grouped", key, g)
        if okey not in self._orders:
            order = torch.empty((lib.sv_conv_tile_order_bytes(n_rows) // 4,), dtype=torch.int32, device=base.device)
            scratch = _lib.workspace.scratch("tile_order", lib.sv_conv_tile_order_scratch_bytes(n_rows), base.device)
            _lib.check(lib.sv_conv_tile_order_grouped(_lib.ptr(masks), _lib.ptr(perm), n_rows, g, _lib.ptr(scratch), _lib.ptr(order), _lib.stream()),
                       "sv_conv_tile_order_grouped")
            self._orders[okey] = (order, base)
        return base, self._orders[okey][0], perm, (direction == "bwd" and self.subm)

    def pair_counts(self):
        lib = _lib.load()
        counts = torch.empty((self.K,), dtype=torch.int32, device=self.nbr_out.device)
        _lib.check(lib.sv_rulebook_pair_counts(_lib.ptr(self.nbr_out), self.n_out, self.K, _lib.ptr(counts), _lib.stream()),
                   "sv_rulebook_pair_counts")
        return counts


# dense cell -> row maps (4 B per cell) up to this size replace the rank dictionary in submanifold rulebooks (MI355X: 288 GB of HBM)
CELLMAP_MAX_BYTES = int(os.environ.get("SEEVCN_CELLMAP_MAX_BYTES", 24 << 30))
TILE_ORDER = os.environ.get("SEEVCN_TILE_ORDER", "1") != "0"      # work-balanced tile order (sv_conv_tile_order); 0: tiles by position
# Group table rows by neighbour mask before the MFMA gather-GEMM (Rulebook.plan): 16-row tiles of equal mask waste almost none of
# their MFMA steps.  Applied to the layers with >= 64 channels on one side (the narrow ones are load-bound and lose more to the
# permuted table reads than they gain); same-box A/B on the bench: GPU time 7.81 -> 7.63 ms per step, the 64->64 kernel 177 -> 145 us.
GROUP_ROWS = os.environ.get("SEEVCN_GROUP_ROWS", "1") != "0"


def build_subm_rulebook(indices, batch_size, spatial_shape, ksize, dilation=(1, 1, 1)):
    lib = _lib.load()
    _lib.require_cuda(indices)
    assert indices.dtype == torch.int32 and indices.dim() == 2 and indices.shape[1] == 4
    indices = indices.contiguous()
    n = indices.shape[0]
    dev = indices.device
    K = int(ksize[0]) * int(ksize[1]) * int(ksize[2])
    ncells = int(batch_size) * int(spatial_shape[0]) * int(spatial_shape[1]) * int(spatial_shape[2])
    nbr = torch.empty((K, n), dtype=torch.int32, device=dev)
    map_bytes = lib.sv_cellmap_persistent_bytes(int(batch_size), _i3(spatial_shape))
    if map_bytes <= CELLMAP_MAX_BYTES:
        cellmap = _lib.workspace.persistent(f"rb_cellmap_{tuple(spatial_shape)}_{batch_size}", map_bytes, dev)
        rc = lib.sv_rulebook_subm_cellmap(_lib.ptr(indices), n, int(batch_size), _i3(spatial_shape), _i3(ksize), _i3(dilation),
                                          _lib.ptr(cellmap), _lib.ptr(nbr), _lib.stream())
        _lib.check(rc, "sv_rulebook_subm_cellmap")
        return Rulebook(nbr, None, indices, list(spatial_shape), n, n, True, list(ksize))
    ws = _lib.workspace.persistent(f"rb_index_{tuple(spatial_shape)}_{batch_size}", lib.sv_index_persistent_bytes(ncells), dev)
    scratch = _lib.workspace.scratch("rb_scratch", lib.sv_rulebook_scratch_bytes(n, ncells), dev)
    rc = lib.sv_rulebook_subm(_lib.ptr(indices), n, int(batch_size), _i3(spatial_shape), _i3(ksize), _i3(dilation),
                              _lib.ptr(ws), _lib.ptr(scratch), _lib.ptr(nbr), _lib.stream())
    _lib.check(rc, "sv_rulebook_subm")
    return Rulebook(nbr, None, indices, list(spatial_shape), n, n, True, list(ksize))


def build_sparse_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation=(1, 1, 1)):
    lib = _lib.load()
    _lib.require_cuda(indices)
    assert indices.dtype == torch.int32 and indices.dim() == 2 and indices.shape[1] == 4
    indices = indices.contiguous()
    n_in = indices.shape[0]
    dev = indices.device
    K = int(ksize[0]) * int(ksize[1]) * int(ksize[2])
    oshape = conv_out_shape(spatial_shape, ksize, stride, padding, dilation)
    ncells = int(batch_size) * oshape[0] * oshape[1] * oshape[2]
    ws = _lib.workspace.persistent(f"rb_index_{tuple(oshape)}_{batch_size}", lib.sv_index_persistent_bytes(ncells), dev)
    scratch = _lib.workspace.scratch("rb_scratch", lib.sv_rulebook_scratch_bytes(n_in, ncells), dev)
    # an input reaches at most prod(ceil(k/s)) outputs; never more than the number of cells
    per_in = 1
    for k, s in zip(ksize, stride):
        per_in *= -(-int(k) // int(s))
    cap = max(min(n_in * per_in, ncells), 1)
    out_coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    nbr_in = torch.empty((K, n_in), dtype=torch.int32, device=dev)
    num_out = torch.zeros((1,), dtype=torch.int32, device=dev)
    rc = lib.sv_rulebook_sparse(_lib.ptr(indices), n_in, int(batch_size), _i3(spatial_shape), _i3(ksize), _i3(stride),
                                _i3(padding), _i3(dilation), _lib.ptr(ws), _lib.ptr(scratch), _lib.ptr(out_coords),
                                _lib.ptr(nbr_in) if n_in else None, cap, _lib.ptr(num_out), _lib.stream())
    _lib.check(rc, "sv_rulebook_sparse")
    n_out = int(num_out.item())  # host needs the size to allocate the output rows (spconv syncs here too)
    out_coords = out_coords[:n_out]
    nbr_out = torch.empty((K, n_out), dtype=torch.int32, device=dev)
    rc = lib.sv_rulebook_invert(_lib.ptr(nbr_in) if n_in else None, n_in, K, _lib.ptr(nbr_out) if n_out else None, n_out, _lib.stream())
    _lib.check(rc, "sv_rulebook_invert")
    return Rulebook(nbr_out, nbr_in, out_coords, oshape, n_in, n_out, False, list(ksize))


def gather_gemm(x, nbr, wt, n_rows, bias=None, scale=None, shift=None, residual=None, relu=False, tile_order=None, row_perm=None,
                table_k_reversed=False):
    """Y (n_rows, Nc) = epi(sum_k X[nbr[k]] @ wt[k].T); wt is ANY (K, Nc, Kd) float32 view (its strides go to the kernel: no
    transposing copy of the parameter per call).  tile_order: Rulebook.tile_order(nbr)."""
    lib = _lib.load()
    K, Nc, Kd = wt.shape
    assert x.shape[1] == Kd and nbr.shape[0] == K and wt.dtype == torch.float32
    x = x.contiguous()
    y = torch.empty((n_rows, Nc), dtype=torch.float32, device=x.device)
    sk, sn, sc = wt.stride()
    rc = lib.sv_sparse_conv_gather_gemm_strided(_lib.ptr(x) if x.numel() else None, x.shape[0], _lib.ptr(nbr) if nbr.numel() else None, ctypes.c_void_p(wt.data_ptr()),
                                                sk, sn, sc, _lib.ptr(y) if n_rows else None, n_rows, K, Kd, Nc, _lib.ptr(bias), _lib.ptr(scale),
                                                _lib.ptr(shift), _lib.ptr(residual), int(bool(relu)), _lib.ptr(tile_order), _lib.ptr(row_perm), int(bool(table_k_reversed)), _lib.stream())
    _lib.check(rc, "sv_sparse_conv_gather_gemm")
    return y


def wgrad(x, nbr, dy, K, cin, cout):
    lib = _lib.load()
    n_rows = dy.shape[0]
    dw = torch.empty((K, cin, cout), dtype=torch.float32, device=dy.device)
    scratch = _lib.workspace.scratch("wgrad", lib.sv_sparse_conv_wgrad_scratch_bytes(n_rows, K, cin, cout), dy.device)
    rc = lib.sv_sparse_conv_wgrad(_lib.ptr(x) if x.numel() else None, _lib.ptr(nbr) if nbr.numel() else None,
                                  _lib.ptr(dy) if n_rows else None, _lib.ptr(dw), n_rows, K, cin, cout, _lib.ptr(scratch), _lib.stream())
    _lib.check(rc, "sv_sparse_conv_wgrad")
    return dw


class SparseConvFunction(torch.autograd.Function):
    """features (N_in,C_in), weight_kio (K,C_in,C_out) -> (N_out,C_out). Backward: gather-GEMM over the input-major
    table for the data gradient and a deterministic row reduction for the weight gradient."""

    @staticmethod
    def forward(ctx, features, weight_kio, rulebook):
        _lib.require_cuda(features, weight_kio)
        features = features.contiguous().float()
        wt = weight_kio.detach().permute(0, 2, 1)               # (K, C_out, C_in) view
        table, order, perm, rev = rulebook.plan("fwd", wt.shape[2], wt.shape[1])
        out = gather_gemm(features, table, wt, rulebook.n_out, tile_order=order, row_perm=perm, table_k_reversed=rev)
        ctx.rulebook = rulebook
        ctx.save_for_backward(features, weight_kio)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        features, weight_kio = ctx.saved_tensors
        rb = ctx.rulebook
        grad_out = grad_out.contiguous().float()
        K, cin, cout = weight_kio.shape
        gf = gw = None
        if ctx.needs_input_grad[0]:
            # dX[i] = sum_k dY[nbr_in[k][i]] @ W[k]^T  -> Wt[k][n=c_in][c=c_out] = W[k][c_in][c_out]: weight_kio itself
            table, order, perm, rev = rb.plan("bwd", cout, cin)
            gf = gather_gemm(grad_out, table, weight_kio.detach(), rb.n_in, tile_order=order, row_perm=perm, table_k_reversed=rev)
        if ctx.needs_input_grad[1]:
            gw = wgrad(features, rb.nbr_out, grad_out, K, cin, cout)
        return gf, gw, None


class DenseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices, batch_size, spatial_shape):
        lib = _lib.load()
        _lib.require_cuda(features, indices)
        features = features.contiguous().float()
        indices = indices.contiguous()
        n, c = features.shape
        d, h, w = (int(s) for s in spatial_shape)
        dev = features.device
        scratch = _lib.workspace.scratch("dense_map", lib.sv_sparse_to_dense_scratch_bytes(batch_size, d, h, w), dev)
        out = torch.empty((batch_size, c, d, h, w), dtype=torch.float32, device=dev)
        rc = lib.sv_sparse_to_dense(_lib.ptr(features) if n else None, _lib.ptr(indices) if n else None, n, batch_size, c, d, h, w,
                                    _lib.ptr(scratch), _lib.ptr(out), _lib.stream())
        _lib.check(rc, "sv_sparse_to_dense")
        ctx.save_for_backward(indices)
        ctx.dims = (batch_size, c, d, h, w)
        return out

    @staticmethod
    def backward(ctx, grad):
        lib = _lib.load()
        (indices,) = ctx.saved_tensors
        b, c, d, h, w = ctx.dims
        n = indices.shape[0]
        grad = grad.contiguous().float()
        out = torch.empty((n, c), dtype=torch.float32, device=grad.device)
        rc = lib.sv_dense_to_sparse(_lib.ptr(grad), _lib.ptr(indices) if n else None, n, b, c, d, h, w, _lib.ptr(out) if n else None,
                                    _lib.stream())
        _lib.check(rc, "sv_dense_to_sparse")
        return out, None, None, None


def sparse_to_dense(features, indices, batch_size, spatial_shape):
    return DenseFunction.apply(features, indices, int(batch_size), list(spatial_shape))
