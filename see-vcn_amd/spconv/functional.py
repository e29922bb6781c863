"""Host wrappers + autograd Functions over the rulebook / sparse-conv entry points of libseevcn_hip.so."""
import ctypes
import math
import os
import weakref

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


class _ArenaViews:
    """Tensors that are slices of one int32 arena, materialised on first access (a torch view costs ~3 us; build_network_index hands out ~80 per
    step and the launch-list chain only ever needs their ADDRESSES).  _lazy: name -> (offset in int32, count, shape or None); addr(name) is the
    device address without making the view, whether the attribute is lazy or a plain tensor."""
    _arena = None
    _lazy = None

    def __getattr__(self, name):
        lazy = self.__dict__.get("_lazy")
        if lazy is not None and name in lazy:
            off, count, shape = lazy[name]
            t = self._arena[off:off + count]
            if shape is not None:
                t = t.view(*shape)
            setattr(self, name, t)
            return t
        raise AttributeError(name)

    def addr(self, name):
        lazy = self.__dict__.get("_lazy")
        if lazy is not None and name in lazy and name not in self.__dict__:
            return self._arena_ptr + 4 * lazy[name][0]
        t = getattr(self, name)
        return 0 if t is None else t.data_ptr()

    def _set_lazy(self, arena, arena_ptr, specs):
        self._arena, self._arena_ptr, self._lazy = arena, arena_ptr, specs


class TablePlan(_ArenaViews):
    """Plan of one rulebook table for the MFMA kernel: the row-major table + masks (from the rulebook builder, or made here from a
    k-major table), the regrouping of the rows (sv_conv_plan_build) and the tile -> wave assignment per tiles-per-wave value
    (sv_conv_plan_tiles).  Built once per table, reused by every launch on it."""

    def __init__(self, table, n_rows, K, rows=None, masks=None, g=None):
        lib = _lib.load()
        dev = table.device
        self.n_rows, self.K = int(n_rows), int(K)
        self._source = table                     # keep the k-major table alive with its plan
        if rows is None or masks is None:
            rows = torch.empty((max(self.n_rows, 1), 32), dtype=torch.int32, device=dev)
            masks = torch.empty((max(self.n_rows, 1),), dtype=torch.int32, device=dev)
            _lib.check(lib.sv_conv_table_rows(_lib.ptr(table), self.n_rows, self.K, _lib.ptr(rows), _lib.ptr(masks), _lib.stream()), "sv_conv_table_rows")
        self.rows, self.masks = rows, masks
        n_perm = lib.sv_conv_plan_perm_bytes(self.n_rows) // 4
        self.perm = torch.empty((n_perm,), dtype=torch.int32, device=dev)
        self.masks_p = torch.empty((n_perm,), dtype=torch.int32, device=dev)
        self._tiles = {}
        if g is not None and FUSED_PLAN:
            # regrouping + the tile deal for this tiles-per-wave value in one launch (one workgroup per region)
            t = torch.empty((lib.sv_conv_plan_tiles_bytes(self.n_rows, int(g)) // 4,), dtype=torch.int32, device=dev)
            _lib.check(lib.sv_conv_plan_build_dealt(_lib.ptr(masks), self.n_rows, int(g), _lib.ptr(self.perm), _lib.ptr(self.masks_p), _lib.ptr(t),
                                                    _lib.stream()), "sv_conv_plan_build_dealt")
            self._tiles[int(g)] = t
            return
        hist = _lib.workspace.persistent("conv_plan_hist", lib.sv_conv_plan_persistent_bytes(), dev)
        _lib.check(lib.sv_conv_plan_build(_lib.ptr(masks), self.n_rows, _lib.ptr(hist), _lib.ptr(self.perm), _lib.ptr(self.masks_p), _lib.stream()),
                   "sv_conv_plan_build")

    @classmethod
    def from_parts(cls, table, n_rows, K, rows, masks, perm, masks_p, g, tile_of):
        """A plan whose pieces were made elsewhere (build_network_index: all plans of a network in one launch)."""
        self = cls.__new__(cls)
        self.n_rows, self.K, self._source = int(n_rows), int(K), table
        self.rows, self.masks, self.perm, self.masks_p = rows, masks, perm, masks_p
        self._tiles = {int(g): tile_of}
        return self

    @classmethod
    def from_arena(cls, arena, arena_ptr, n_rows, K, table_name, owner, rows, masks, perm, masks_p, g, tile_of):
        """from_parts with every piece given as an (offset, count, shape) slice of `arena`; `owner`.`table_name` is the k-major table."""
        self = cls.__new__(cls)
        self.n_rows, self.K = int(n_rows), int(K)
        self._owner, self._table_name = weakref.ref(owner), table_name
        self._tiles = {}
        self._tiles_lazy = {int(g): tile_of}
        self._set_lazy(arena, arena_ptr, {"rows": rows, "masks": masks, "perm": perm, "masks_p": masks_p})
        return self

    @property
    def source(self):
        src = self.__dict__.get("_source")
        if src is not None:
            return src
        owner = self._owner()                      # a WEAK reference: the rulebook holds its plans, a strong one back made a cycle that kept the whole
        if owner is None:                          # index arena of a batch (~450 MB) alive until the cyclic collector ran
            raise RuntimeError("the rulebook of this plan is gone")
        return getattr(owner, self._table_name)

    def tiles_addr(self, g):
        lz = self.__dict__.get("_tiles_lazy")
        if lz is not None and int(g) in lz and int(g) not in self._tiles:
            return self._arena_ptr + 4 * lz[int(g)][0]
        return self.tiles(g).data_ptr()

    def tiles(self, g):
        g = int(g)
        lz = self.__dict__.get("_tiles_lazy")
        if g not in self._tiles and lz is not None and g in lz:
            off, count, _ = lz[g]
            self._tiles[g] = self._arena[off:off + count]
        if g not in self._tiles:
            lib = _lib.load()
            t = torch.empty((lib.sv_conv_plan_tiles_bytes(self.n_rows, g) // 4,), dtype=torch.int32, device=self.perm.device)
            _lib.check(lib.sv_conv_plan_tiles(_lib.ptr(self.masks_p), self.n_rows, g, _lib.ptr(t), _lib.stream()), "sv_conv_plan_tiles")
            self._tiles[g] = t
        return self._tiles[g]


class Rulebook(_ArenaViews):
    """Output-major table nbr_out (K, N_out) plus, for strided convs, the input-major nbr_in (K, N_in)."""

    def __init__(self, nbr_out, nbr_in, out_indices, out_shape, n_in, n_out, subm, ksize):
        self.nbr_out, self.nbr_in = nbr_out, nbr_in
        self.out_indices, self.out_shape = out_indices, out_shape
        self.n_in, self.n_out, self.subm, self.ksize = n_in, n_out, subm, ksize
        self._nbr_in_subm = None
        self.rows_out = self.masks_out = self.rows_in = self.masks_in = None     # row-major twins + neighbour masks from the builders
        self._plans = {}
        self._plan_results = {}
        self.in_indices = self.in_shape = None      # set by the conv that built the rulebook

    @property
    def K(self):
        k = self.__dict__.get("_K")
        return k if k is not None else self.nbr_out.shape[0]

    def table_for_backward_data(self):
        """Input-major k-major table (the plain kernels). For SubM it is the output-major table with the offsets reversed
        (coord[j] = coord[i] + d  <=>  coord[i] = coord[j] - d)."""
        if not self.subm:
            return self.nbr_in
        if self._nbr_in_subm is None:
            self._nbr_in_subm = torch.flip(self.nbr_out, dims=[0]).contiguous()
        return self._nbr_in_subm

    def plan(self, direction, kd, nc):
        """(TablePlan, tile_of, tiles_per_wave, table_k_reversed) for a (kd -> nc)-channel gather-GEMM over this rulebook, or None when the plan
        kernel does not take the layer (sv_conv_mfma_kernel_applies is the single source of truth; SEEVCN_SPCONV_PLAN=0 forces the plain
        kernels for A/B runs).  direction 'fwd' = output-major table, 'bwd' = input-major table; a submanifold table serves its own data
        gradient with the offsets read in reverse, so it has one plan."""
        hit = self._plan_results.get((direction, kd, nc, USE_PLAN))
        if hit is not None:
            return hit[0]
        out = self._plan_uncached(direction, kd, nc)
        self._plan_results[(direction, kd, nc, USE_PLAN)] = (out,)          # asked 60 times per step; two ctypes calls each before
        return out

    def _plan_uncached(self, direction, kd, nc):
        assert direction in ("fwd", "bwd")
        kd, nc = int(kd), int(nc)
        n_rows = self.n_out if direction == "fwd" else self.n_in
        n_src = self.n_in if direction == "fwd" else self.n_out
        lib = _lib.load()
        if not USE_PLAN or n_rows == 0 or not lib.sv_conv_mfma_kernel_applies(int(self.K), kd, nc, int(n_src)):
            return None
        key = "fwd" if (direction == "fwd" or self.subm) else "bwd"
        g = lib.sv_conv_tiles_per_wave(n_rows, kd, nc)
        if key not in self._plans:
            if key == "fwd":
                self._plans[key] = TablePlan(self.nbr_out, n_rows, self.K, self.rows_out, self.masks_out, g=g)
            else:
                self._plans[key] = TablePlan(self.nbr_in, n_rows, self.K, self.rows_in, self.masks_in, g=g)
        tp = self._plans[key]
        return tp, tp.tiles(g), g, (direction == "bwd" and self.subm)

    def plan_addrs(self, direction, kd, nc):
        """plan() as device addresses for a launch list: (rows, perm, masks_p, tile_of, tiles_per_wave, table_k_reversed) or None -- no tensor view
        is made for pieces that live in an arena (build_network_index)."""
        key = ("addr", direction, kd, nc, USE_PLAN)
        hit = self._plan_results.get(key)
        if hit is not None:
            return hit[0]
        out = None
        if USE_PLAN:
            pk = "fwd" if (direction == "fwd" or self.subm) else "bwd"
            tp = self._plans.get(pk)
            n_rows = self.n_out if direction == "fwd" else self.n_in
            if tp is not None and n_rows > 0:
                lib = _lib.load()
                n_src = self.n_in if direction == "fwd" else self.n_out
                if lib.sv_conv_mfma_kernel_applies(int(self.K), int(kd), int(nc), int(n_src)):
                    g = lib.sv_conv_tiles_per_wave(n_rows, int(kd), int(nc))
                    out = (tp.addr("rows"), tp.addr("perm"), tp.addr("masks_p"), tp.tiles_addr(g), g, (direction == "bwd" and self.subm))
            else:
                p = self.plan(direction, kd, nc)
                if p is not None:
                    tp, tile_of, g, rev = p
                    out = (tp.addr("rows"), tp.addr("perm"), tp.addr("masks_p"), tile_of.data_ptr(), g, rev)
        self._plan_results[key] = (out,)
        return out

    def wgrad_plan(self, cin, cout):
        """Device plan of the equal-pieces weight gradient (sv_wgrad_plan_build) for a (cin -> cout) layer on this table, or None when that kernel does
        not take the layer (sv_wgrad_planned_applies; SEEVCN_WGRAD_PLANNED=0 forces the chunked kernel for A/B runs).  One plan per table and piece
        count: the layers that share an indice_key share it."""
        if not WGRAD_PLANNED or self.n_out == 0:
            return None
        key = ("wgrad", int(cin), int(cout))
        hit = self._plan_results.get(key)
        if hit is not None:
            return hit[0]
        lib = _lib.load()
        out = None
        if lib.sv_wgrad_planned_applies(int(self.n_in), int(self.n_out), int(self.K), int(cin), int(cout)):
            pieces = lib.sv_wgrad_plan_pieces(int(cin), int(cout))
            plans = self.__dict__.setdefault("_wgrad_plans", {})
            if pieces not in plans:
                nbr = self.nbr_out
                buf = torch.empty((lib.sv_wgrad_plan_bytes(self.n_out, int(self.K), pieces),), dtype=torch.uint8, device=nbr.device)
                _lib.check(lib.sv_wgrad_plan_build(_lib.ptr(nbr), self.n_out, int(self.K), pieces, buf.data_ptr(), _lib.stream()), "sv_wgrad_plan_build")
                plans[pieces] = buf
            out = plans[pieces]
        self._plan_results[key] = (out,)
        return out

    def pair_counts(self):
        lib = _lib.load()
        counts = torch.empty((self.K,), dtype=torch.int32, device=self.nbr_out.device)
        _lib.check(lib.sv_rulebook_pair_counts(_lib.ptr(self.nbr_out), self.n_out, self.K, _lib.ptr(counts), _lib.stream()),
                   "sv_rulebook_pair_counts")
        return counts


# dense cell -> row maps (4 B per cell) up to this size replace the rank dictionary in submanifold rulebooks (MI355X: 288 GB of HBM)
CELLMAP_MAX_BYTES = int(os.environ.get("SEEVCN_CELLMAP_MAX_BYTES", 24 << 30))
FUSED_PLAN = os.environ.get("SEEVCN_FUSED_PLAN", "1") != "0"     # 0: plans built by the four separate kernels (A/B runs, tests)
USE_PLAN = os.environ.get("SEEVCN_SPCONV_PLAN", "1") != "0"      # 0: every layer on the plain kernels (A/B runs, tests)
WGRAD_PLANNED = os.environ.get("SEEVCN_WGRAD_PLANNED", "1") != "0"   # 0: weight gradients on (row chunk, offset) workgroups instead of equal pieces (A/B runs, tests)


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
        # keyed by the grid only: the map is b-major, so one sized for the largest batch seen serves every smaller batch (Workspace grows it)
        cellmap = _lib.workspace.persistent(f"rb_cellmap_{tuple(spatial_shape)}", map_bytes, dev)
        with_rows = K <= 27 and n > 0
        rows = torch.empty((n, 32), dtype=torch.int32, device=dev) if with_rows else None
        masks = torch.empty((n,), dtype=torch.int32, device=dev) if with_rows else None
        rc = lib.sv_rulebook_subm_cellmap(_lib.ptr(indices), n, int(batch_size), _i3(spatial_shape), _i3(ksize), _i3(dilation),
                                          _lib.ptr(cellmap), _lib.ptr(nbr), _lib.ptr(rows), _lib.ptr(masks), _lib.stream())
        _lib.check(rc, "sv_rulebook_subm_cellmap")
        rb = Rulebook(nbr, None, indices, list(spatial_shape), n, n, True, list(ksize))
        rb.rows_out, rb.masks_out = rows, masks
        return rb
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
    num_out = torch.empty((1,), dtype=torch.int32, device=dev)      # zeroed by sv_rulebook_sparse
    with_rows = K <= 27 and n_in > 0
    in_block = torch.empty((33 * n_in,), dtype=torch.int32, device=dev) if with_rows else None     # [rows (n_in, 32) | masks (n_in)]
    rc = lib.sv_rulebook_sparse(_lib.ptr(indices), n_in, int(batch_size), _i3(spatial_shape), _i3(ksize), _i3(stride),
                                _i3(padding), _i3(dilation), _lib.ptr(ws), _lib.ptr(scratch), _lib.ptr(out_coords),
                                _lib.ptr(nbr_in) if n_in else None, _lib.ptr(in_block), cap, _lib.ptr(num_out), _lib.stream())
    _lib.check(rc, "sv_rulebook_sparse")
    n_out = _lib.host_int(num_out)  # host needs the size to allocate the output rows (spconv syncs here too)
    out_coords = out_coords[:n_out]
    if not with_rows or n_out == 0:
        nbr_out = torch.empty((K, n_out), dtype=torch.int32, device=dev)
        rc = lib.sv_rulebook_invert(_lib.ptr(nbr_in) if n_in else None, n_in, K, _lib.ptr(nbr_out) if n_out else None, n_out, _lib.stream())
        _lib.check(rc, "sv_rulebook_invert")
        return Rulebook(nbr_out, nbr_in, out_coords, oshape, n_in, n_out, False, list(ksize))
    # one block for the output side: [rows (n_out, 32) | k-major table (K, n_out) | masks (n_out)] -- the row-major twins and masks feed the plans
    out_block = torch.empty(((32 + K + 1) * n_out,), dtype=torch.int32, device=dev)
    rc = lib.sv_rulebook_invert_rows(_lib.ptr(in_block), n_in, K, _lib.ptr(out_block), n_out, _lib.stream())
    _lib.check(rc, "sv_rulebook_invert_rows")
    nbr_out = out_block[32 * n_out:(32 + K) * n_out].view(K, n_out)
    rb = Rulebook(nbr_out, nbr_in, out_coords, oshape, n_in, n_out, False, list(ksize))
    rb.rows_out, rb.masks_out = out_block[:32 * n_out].view(n_out, 32), out_block[(32 + K) * n_out:]
    rb.rows_in, rb.masks_in = in_block[:32 * n_in].view(n_in, 32), in_block[32 * n_in:]
    return rb


class ConvSpec:
    """What build_network_index needs to know about one sparse convolution, in execution order."""
    __slots__ = ("key", "subm", "ksize", "stride", "padding", "dilation", "cin", "cout")

    def __init__(self, key, subm, ksize, stride, padding, dilation, cin, cout):
        self.key, self.subm = key, bool(subm)
        self.ksize, self.stride, self.padding, self.dilation = ([int(v) for v in t] for t in (ksize, stride, padding, dilation))
        self.cin, self.cout = int(cin), int(cout)


def _al(n, q=64):
    return (int(n) + q - 1) // q * q


def build_network_index(coords, batch_size, spatial_shape, specs, n0_dev=None, with_backward=True):
    """Every rulebook AND conv plan of a network in 12 + 4 launches and ONE device -> host read (round 3: ~57 launches, one read per strided
    level): sv_rulebook_chain_count walks the strided levels end to end on the device (level l + 1 marks from level l's freshly written site
    list), the site counts -- and, with n0_dev, the voxel count the caller has not read yet -- come back together, then sv_rulebook_batch fills
    every table through the levels' dense cell maps (SET / QUERY / CLEAR, one launch each) and sv_conv_plan_build_dealt_batch makes all plans in
    one launch.  specs: ConvSpec per convolution in execution order; a strided spec moves to the next level, submanifold specs of one key
    share a table.  coords (cap0, 4) int32 [b,z,y,x]: the first n0 rows count (n0 = *n0_dev or cap0).
    -> (n0, {key: Rulebook}) with tables bit-identical to build_subm_rulebook / build_sparse_rulebook, or None (nothing launched) when the
    network does not fit the batch form: a kernel wider than 3, a cell map beyond CELLMAP_MAX_BYTES, more than 8 levels, no rows."""
    import numpy as np
    lib = _lib.load()
    _lib.require_cuda(coords)
    assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
    coords = coords.contiguous()
    dev, B, cap0 = coords.device, int(batch_size), int(coords.shape[0])
    strided = [sp for sp in specs if not sp.subm]
    if cap0 == 0 or not specs or len(strided) > 8 or any(max(sp.ksize) > 3 for sp in specs) or any(not (k & 1) for sp in specs if sp.subm for k in sp.ksize):
        return None
    shapes = [[int(v) for v in spatial_shape]]
    for sp in strided:
        shapes.append(conv_out_shape(shapes[-1], sp.ksize, sp.stride, sp.padding, sp.dilation))
    L = len(strided)
    map_bytes = [lib.sv_cellmap_persistent_bytes(B, _i3(sh)) for sh in shapes]
    if max(map_bytes) > CELLMAP_MAX_BYTES:
        return None
    ncells = [B * sh[0] * sh[1] * sh[2] for sh in shapes]
    # ---- phase 1: the site sets of all strided levels, counted on the device
    counts = None
    if L:
        caps, cap = [], cap0
        for l, sp in enumerate(strided):
            per_in = 1
            for k, st, dl in zip(sp.ksize, sp.stride, sp.dilation):
                # per axis an input reaches the outputs o with o * st = x + pad - j * dl, j < k: ceil(k / st) of them when dilation and stride are
                # coprime (the residues j * dl mod st then cycle evenly), up to k when they share a factor (dilation 2, stride 2: every j fits)
                per_in *= -(-k // st) if math.gcd(dl, st) == 1 else k
            cap = max(min(cap * per_in, ncells[l + 1]), 1)           # never more than the cells
            caps.append(cap)
        works = [_lib.workspace.persistent(f"rb_index_{tuple(shapes[l + 1])}_{B}", lib.sv_index_persistent_bytes(ncells[l + 1]), dev) for l in range(L)]
        assert len({w.data_ptr() for w in works}) == L, "two levels of a chain on one grid would share their index"
        sites = [_lib.workspace.scratch(f"rb_chain_sites_{l}", caps[l] * 16, dev) for l in range(L)]
        scratch = _lib.workspace.scratch("rb_chain_scratch", lib.sv_rulebook_chain_scratch_bytes(max(ncells[1:])), dev)
        num_out = torch.empty((L,), dtype=torch.int32, device=dev)
        geoms = []
        for l, sp in enumerate(strided):
            geoms += shapes[l] + sp.ksize + sp.stride + sp.padding + sp.dilation
        rc = lib.sv_rulebook_chain_count(_lib.ptr(coords), cap0, _lib.ptr(n0_dev), B, L, _lib.host_array(ctypes.c_int32, geoms),
                                         _lib.host_array(ctypes.c_void_p, [w.data_ptr() for w in works]),
                                         _lib.host_array(ctypes.c_void_p, [t.data_ptr() for t in sites]), _lib.host_array(ctypes.c_int64, caps),
                                         _lib.ptr(num_out), _lib.ptr(scratch), _lib.stream())
        _lib.check(rc, "sv_rulebook_chain_count")
    try:
        if L:
            vals = _lib.host_ints(([n0_dev] if n0_dev is not None else []) + [num_out])        # THE read
        else:
            vals = _lib.host_ints([n0_dev]) if n0_dev is not None else []
        n = [vals.pop(0) if n0_dev is not None else cap0] + vals
        if any(n[l + 1] > caps[l] for l in range(L)):                  # not an assert: `python -O` must not turn a truncated site list into wrong tables
            raise _lib.SeevcnHipError(f"build_network_index: a strided level emitted {n[1:]} output sites for capacities {caps}")
    except BaseException:
        if L:
            for w in works:                                            # phase 1 left its marks: a failed read must not leak them into the next call
                w.zero_()
        raise
    if n[0] != cap0:
        coords = coords[:n[0]]
    # ---- layout of ONE int32 arena for every table and plan
    total = [0]

    def take(count):
        off = total[0]
        total[0] += _al(max(int(count), 0))
        return off, int(count)

    level_of, lv, tables, plans = {}, 0, {}, []                         # tables: key -> dict of arena slices
    for sp in specs:
        K = sp.ksize[0] * sp.ksize[1] * sp.ksize[2]
        if sp.subm:
            if sp.key not in tables:
                tables[sp.key] = dict(spec=sp, level=lv, K=K, nbr=take(K * n[lv]), rows=take(32 * n[lv]), masks=take(n[lv]), plans={})
            t = tables[sp.key]
            assert t["spec"].subm and t["spec"].ksize == sp.ksize and t["level"] == lv, f"indice_key {sp.key} reused with a different kernel or level"
        else:
            assert sp.key not in tables, f"indice_key {sp.key} used by two strided convolutions"
            tables[sp.key] = dict(spec=sp, level=lv, K=K, out_idx=take(4 * n[lv + 1]), nbr_in=take(K * n[lv]), in_block=take(33 * n[lv]),
                                  out_block=take((32 + K + 1) * n[lv + 1]), plans={})
            t = tables[sp.key]
        # plans this convolution will ask for (Rulebook.plan): forward on the output-major table, data gradient on the input-major one (a
        # submanifold table serves both)
        n_in_rows, n_out_rows = n[t["level"]], n[t["level"] + (0 if sp.subm else 1)]
        wants = [("fwd", n_out_rows, n_in_rows, sp.cin, sp.cout)]
        if with_backward:
            wants.append(("fwd" if sp.subm else "bwd", n_in_rows, n_out_rows, sp.cout, sp.cin))
        for pkey, n_rows, n_src, kd, nc in wants:
            if USE_PLAN and FUSED_PLAN and n_rows > 0 and pkey not in t["plans"] and lib.sv_conv_mfma_kernel_applies(K, kd, nc, n_src):
                g = lib.sv_conv_tiles_per_wave(n_rows, kd, nc)
                n_perm = lib.sv_conv_plan_perm_bytes(n_rows) // 4
                t["plans"][pkey] = dict(n_rows=n_rows, g=g, perm=take(n_perm), masks_p=take(n_perm), tiles=take(lib.sv_conv_plan_tiles_bytes(n_rows, g) // 4))
        if not sp.subm:
            lv += 1
    arena = torch.empty((max(total[0], 1),), dtype=torch.int32, device=dev)
    base = arena.data_ptr()

    def view(sl, *shape):
        return arena[sl[0]:sl[0] + sl[1]].view(*shape) if shape else arena[sl[0]:sl[0] + sl[1]]

    def addr(sl, extra=0):
        return base + 4 * (sl[0] + extra)

    # ---- phase 2: SET every level, QUERY every table, CLEAR the maps
    maps = [_lib.workspace.persistent(f"rb_cellmap_{tuple(sh)}", mb, dev) for sh, mb in zip(shapes, map_bytes)]
    strided_tabs = [tables[sp.key] for sp in strided]
    jobs = []

    def row(*vals):
        r = [0] * 32
        r[:len(vals)] = [int(v) for v in vals]
        jobs.append(r)

    level_sites = [coords.data_ptr() if n[0] else 0] + [addr(t["out_idx"]) for t in strided_tabs]
    row(1, level_sites[0], 0, maps[0].data_ptr(), 0, n[0], *shapes[0], B)
    for l, t in enumerate(strided_tabs):
        row(1, sites[l].data_ptr(), level_sites[l + 1], maps[l + 1].data_ptr(), works[l].data_ptr(), n[l + 1], *shapes[l + 1], B)
    for t in tables.values():
        sp, l, K = t["spec"], t["level"], t["K"]
        if sp.subm:
            row(2, level_sites[l], maps[l].data_ptr(), addr(t["nbr"]), addr(t["rows"]), addr(t["masks"]), n[l], *shapes[l], *shapes[l], *sp.ksize,
                1, 1, 1, *[-(k // 2) * d for k, d in zip(sp.ksize, sp.dilation)], *sp.dilation, 1, 1, 1, B)
        else:
            no, ni = n[l + 1], n[l]
            # output-major: [rows_out (n_out, 32) | nbr_out (K, n_out) | masks_out (n_out)], rows = the output sites, target = the input level
            row(2, level_sites[l + 1], maps[l].data_ptr(), addr(t["out_block"], 32 * no), addr(t["out_block"]), addr(t["out_block"], (32 + K) * no), no,
                *shapes[l + 1], *shapes[l], *sp.ksize, *sp.stride, *[-p for p in sp.padding], *sp.dilation, 1, 1, 1, B)
            # input-major: nbr_in (K, n_in) + [rows_in (n_in, 32) | masks_in (n_in)], rows = the input sites, target = the output level
            row(2, level_sites[l], maps[l + 1].data_ptr(), addr(t["nbr_in"]), addr(t["in_block"]), addr(t["in_block"], 32 * ni), ni,
                *shapes[l], *shapes[l + 1], *sp.ksize, 1, 1, 1, *sp.padding, *[-d for d in sp.dilation], *sp.stride, B)
    arr = np.array(jobs, dtype=np.int64)
    try:
        _lib.check(lib.sv_rulebook_batch(arr.ctypes.data, len(jobs), _lib.stream()), "sv_rulebook_batch")
    except BaseException:
        for w in (works if L else []):                                 # a refused batch leaves phase 1's marks (and maybe map entries) behind:
            w.zero_()                                                  # the persistent workspaces must be all-zero for the next call
        for m in maps:
            m.zero_()
        raise
    # ---- all plans in one launch
    pj = []
    for t in tables.values():
        sp, l, K = t["spec"], t["level"], t["K"]
        for pkey, pl in t["plans"].items():
            if sp.subm:
                masks = addr(t["masks"])
            elif pkey == "fwd":
                masks = addr(t["out_block"], (32 + K) * n[l + 1])
            else:
                masks = addr(t["in_block"], 32 * n[l])
            pj.append([masks, pl["n_rows"], pl["g"], addr(pl["perm"]), addr(pl["masks_p"]), addr(pl["tiles"]), 0, 0])
    if pj:
        parr = np.array(pj, dtype=np.int64)
        _lib.check(lib.sv_conv_plan_build_dealt_batch(parr.ctypes.data, len(pj), _lib.stream()), "sv_conv_plan_build_dealt_batch")
    # ---- the Rulebook objects: every table a lazy slice of the arena (views are made when something asks for the tensor)
    out = {}
    level_idx = [coords]
    for t in strided_tabs:
        level_idx.append(view(t["out_idx"], n[t["level"] + 1], 4))
    for key, t in tables.items():
        sp, l, K = t["spec"], t["level"], t["K"]
        if sp.subm:
            rb = Rulebook(None, None, level_idx[l], list(shapes[l]), n[l], n[l], True, list(sp.ksize))
            lazy = {"nbr_out": (t["nbr"][0], K * n[l], (K, n[l])), "rows_out": (t["rows"][0], 32 * n[l], (n[l], 32)), "masks_out": (t["masks"][0], n[l], None)}
        else:
            no, ni = n[l + 1], n[l]
            ob, ib = t["out_block"][0], t["in_block"][0]
            rb = Rulebook(None, None, level_idx[l + 1], list(shapes[l + 1]), ni, no, False, list(sp.ksize))
            lazy = {"nbr_out": (ob + 32 * no, K * no, (K, no)), "rows_out": (ob, 32 * no, (no, 32)), "masks_out": (ob + (32 + K) * no, no, None),
                    "nbr_in": (t["nbr_in"][0], K * ni, (K, ni)), "rows_in": (ib, 32 * ni, (ni, 32)), "masks_in": (ib + 32 * ni, ni, None)}
        for name in ("nbr_out", "nbr_in", "rows_out", "masks_out", "rows_in", "masks_in"):
            rb.__dict__.pop(name, None)                    # the constructor's None placeholders: the lazy table takes over
        if sp.subm:
            rb.nbr_in = rb.rows_in = rb.masks_in = None
        rb._K = K
        rb._set_lazy(arena, base, lazy)
        rb.in_indices, rb.in_shape = level_idx[l], list(shapes[l])
        for pkey, pl in t["plans"].items():
            side = "out" if pkey == "fwd" else "in"
            rb._plans[pkey] = TablePlan.from_arena(arena, base, pl["n_rows"], K, "nbr_" + side, rb, lazy["rows_" + side], lazy["masks_" + side],
                                                   (pl["perm"][0], pl["perm"][1], None), (pl["masks_p"][0], pl["masks_p"][1], None), pl["g"],
                                                   (pl["tiles"][0], pl["tiles"][1], None))
        out[key] = rb
    return n[0], out


class _FragmentCache:
    """Weights in MFMA fragment order (sv_conv_weight_fragments), both directions in one launch, into buffers owned by the weight tensor's
    cache entry (tied to the base tensor by a weak reference: a data pointer alone can be handed to a new tensor after the old one is
    freed).  The buffers belong to the entry, not to the library: convs on different streams never share a staging buffer (round 1 kept one
    device-global buffer).  The fragments are RE-LAID ON EVERY FORWARD (one 4 us launch per layer): a tensor's version counter cannot be
    trusted to detect an update -- torch's fused optimisers (SGD(fused=True), the bench's) change the weights without bumping it --
    and the backward of the same iteration takes the pair the forward made (SparseConvFunction keeps it in ctx)."""

    def __init__(self):
        self._d = {}
        self._fresh = set()          # entries re-laid by refresh_all and not yet taken by their layer's forward
        self._tables = {}            # descriptor tables of refresh_all, by the identity of the weight set

    def _entry(self, weight_kio):
        base = weight_kio._base if weight_kio._base is not None else weight_kio
        K, cin, cout = weight_kio.shape
        key = id(base)
        hit = self._d.get(key)
        if hit is not None and hit[0]() is base and hit[1].numel() == K * cin * cout and hit[1].device == weight_kio.device:
            return key, hit[1], hit[2], True
        fwd = torch.empty((K * cin * cout,), dtype=torch.float32, device=weight_kio.device)
        bwd = torch.empty_like(fwd)
        d = self._d
        self._d[key] = (weakref.ref(base, lambda _r, k=key: (d.pop(k, None), self._fresh.discard(k))), fwd, bwd)
        return key, fwd, bwd, False

    def refresh_all(self, weights):
        """Re-lay the fragments of every (K, C_in, C_out) weight view in `weights` in ONE launch (a backbone calls this at the top of its
        forward); each layer's get() of this forward then takes its pair without a launch of its own.  An entry stays fresh only until it is
        taken once or the next refresh_all."""
        lib = _lib.load()
        self._fresh.clear()
        weights = [w for w in weights if w.is_cuda and w.dtype == torch.float32 and w.shape[1] % 16 == 0 and w.shape[2] % 16 == 0]
        if not weights:
            return
        rows, keys, unit0 = [], [], 0
        for w in weights:
            key, fwd, bwd, _ = self._entry(w)
            K, cin, cout = w.shape
            sk, si, so = w.stride()
            rows.append([w.data_ptr(), sk, si, so, K, cin, cout, fwd.data_ptr(), bwd.data_ptr(), unit0])
            unit0 += 2 * K * cin * cout // 4
            keys.append(key)
        sig = tuple(tuple(r) for r in rows)
        table = self._tables.get(sig)
        if table is None:
            if len(self._tables) > 16:
                self._tables.clear()
            table = torch.tensor(rows, dtype=torch.int64).to(weights[0].device)
            self._tables[sig] = table
        _lib.check(lib.sv_conv_weight_fragments_batch(_lib.ptr(table), len(rows), unit0, _lib.stream()), "sv_conv_weight_fragments_batch")
        self._fresh.update(keys)

    def get(self, weight_kio, refresh=True):
        K, cin, cout = weight_kio.shape
        key, fwd, bwd, hit = self._entry(weight_kio)
        if hit and (not refresh or key in self._fresh):
            self._fresh.discard(key)
            return fwd, bwd
        lib = _lib.load()
        sk, si, so = weight_kio.stride()
        _lib.check(lib.sv_conv_weight_fragments(ctypes.c_void_p(weight_kio.data_ptr()), sk, si, so, K, cin, cout, _lib.ptr(fwd), _lib.ptr(bwd), _lib.stream()),
                   "sv_conv_weight_fragments")
        return fwd, bwd

    def clear(self):
        self._d.clear()
        self._fresh.clear()
        self._tables.clear()


fragment_cache = _FragmentCache()


def gather_gemm_planned(x, plan, wfrag, n_rows, K, kd, nc, bias=None, scale=None, shift=None, residual=None, relu=False, bn_partial=None):
    """Y (n_rows, nc) = epi(sum_k X[nbr[k]] @ W[k]) on a table plan (Rulebook.plan) with the weights in fragment order.  bn_partial: device address
    (int) of sv_conv_planned_partials() x 2 x nc floats that receive the BatchNorm partial sums of Y (plain epilogue only)."""
    lib = _lib.load()
    tp, tile_of, g, rev = plan
    assert x.shape[1] == kd and x.dtype == torch.float32
    x = x.contiguous()
    y = torch.empty((n_rows, nc), dtype=torch.float32, device=x.device)
    rc = lib.sv_sparse_conv_gather_gemm_planned(_lib.ptr(x) if x.numel() else None, x.shape[0], _lib.ptr(tp.rows), _lib.ptr(tp.perm), _lib.ptr(tp.masks_p),
                                                _lib.ptr(tile_of), int(g), _lib.ptr(wfrag),
                                                _lib.ptr(y) if n_rows else None, n_rows, int(K), int(kd), int(nc), _lib.ptr(bias), _lib.ptr(scale),
                                                _lib.ptr(shift), _lib.ptr(residual), int(bool(relu)), int(bool(rev)), bn_partial, _lib.stream())
    _lib.check(rc, "sv_sparse_conv_gather_gemm_planned")
    return y


def gather_gemm(x, nbr, wt, n_rows, bias=None, scale=None, shift=None, residual=None, relu=False):
    """Plain kernels: Y (n_rows, Nc) = epi(sum_k X[nbr[k]] @ wt[k].T) with the k-major table and a (K, Nc, Kd) weight (made contiguous here:
    only the 3-channel input layer and odd channel counts come this way)."""
    lib = _lib.load()
    K, Nc, Kd = wt.shape
    assert x.shape[1] == Kd and nbr.shape[0] == K and wt.dtype == torch.float32
    x = x.contiguous()
    wt = wt.contiguous()
    y = torch.empty((n_rows, Nc), dtype=torch.float32, device=x.device)
    rc = lib.sv_sparse_conv_gather_gemm(_lib.ptr(x) if x.numel() else None, x.shape[0], _lib.ptr(nbr) if nbr.numel() else None, _lib.ptr(wt),
                                        _lib.ptr(y) if n_rows else None, n_rows, K, Kd, Nc, _lib.ptr(bias), _lib.ptr(scale),
                                        _lib.ptr(shift), _lib.ptr(residual), int(bool(relu)), _lib.stream())
    _lib.check(rc, "sv_sparse_conv_gather_gemm")
    return y


def wgrad(x, nbr, dy, K, cin, cout, like=None, plan=None):
    """dW (K, C_in, C_out).  `like`: a (K, C_in, C_out) VIEW of the parameter (SparseConvolution.weight_kio()); the gradient is then written
    in that view's memory layout (same strides over a fresh dense buffer), so that autograd's way back through the permute / reshape of the
    view is a view again instead of a transposing copy (one small launch per layer and step otherwise).  `plan`: Rulebook.wgrad_plan(cin, cout) of
    the table -- stage 1 then runs on equal pieces (sv_sparse_conv_wgrad_planned) instead of (row chunk, offset) workgroups."""
    lib = _lib.load()
    n_rows = dy.shape[0]
    xs, ns, ds = (_lib.ptr(x) if x.numel() else None), (_lib.ptr(nbr) if nbr.numel() else None), (_lib.ptr(dy) if n_rows else None)
    st = None if like is None else tuple(int(v) for v in like.stride())
    strided = st is not None and not like.is_contiguous() and min(st) > 0 and sorted(st)[0] == 1 and _dense_permutation(tuple(like.shape), st)
    if strided:
        dw = torch.empty_strided((K, cin, cout), st, dtype=torch.float32, device=dy.device)
    else:
        dw = torch.empty((K, cin, cout), dtype=torch.float32, device=dy.device)
        st = (0, 0, 0)
    if plan is not None:
        partial = _lib.workspace.scratch("wgrad", lib.sv_sparse_conv_wgrad_planned_bytes(K, cin, cout), dy.device)
        rc = lib.sv_sparse_conv_wgrad_planned(xs, int(x.shape[0]), ns, ds, dw.data_ptr(), n_rows, K, cin, cout, st[0], st[1], st[2], plan.data_ptr(),
                                              _lib.ptr(partial), _lib.stream())
        _lib.check(rc, "sv_sparse_conv_wgrad_planned")
        return dw
    scratch = _lib.workspace.scratch("wgrad", lib.sv_sparse_conv_wgrad_scratch_bytes(n_rows, K, cin, cout), dy.device)
    if strided:
        rc = lib.sv_sparse_conv_wgrad_strided(xs, int(x.shape[0]), ns, ds, dw.data_ptr(), n_rows, K, cin, cout, st[0], st[1], st[2], _lib.ptr(scratch), _lib.stream())
        _lib.check(rc, "sv_sparse_conv_wgrad_strided")
        return dw
    rc = lib.sv_sparse_conv_wgrad(xs, int(x.shape[0]), ns, ds, _lib.ptr(dw), n_rows, K, cin, cout, _lib.ptr(scratch), _lib.stream())
    _lib.check(rc, "sv_sparse_conv_wgrad")
    return dw


def build_wgrad_plans(items):
    """Rulebook.wgrad_plan for several (rulebook, cin, cout) at once: the plans that are missing come out of ONE allocation and two launches
    (sv_wgrad_plan_build_batch) instead of two launches per table."""
    import numpy as np
    if not WGRAD_PLANNED:
        return
    lib = _lib.load()
    todo, seen, total = [], set(), 0
    for rb, cin, cout in items:
        key = ("wgrad", int(cin), int(cout))
        if rb.n_out == 0 or key in rb._plan_results:
            continue
        if not lib.sv_wgrad_planned_applies(int(rb.n_in), int(rb.n_out), int(rb.K), int(cin), int(cout)):
            rb._plan_results[key] = (None,)
            continue
        pieces = lib.sv_wgrad_plan_pieces(int(cin), int(cout))
        plans = rb.__dict__.setdefault("_wgrad_plans", {})
        if pieces not in plans and (id(rb), pieces) not in seen:
            seen.add((id(rb), pieces))
            nbytes = lib.sv_wgrad_plan_bytes(rb.n_out, int(rb.K), pieces)
            todo.append((rb, pieces, total, nbytes))
            total += nbytes
    if todo:
        dev = todo[0][0].out_indices.device
        buf = torch.empty((total,), dtype=torch.uint8, device=dev)
        jobs = np.zeros((len(todo), 8), dtype=np.int64)
        for q, (rb, pieces, off, nbytes) in enumerate(todo):
            jobs[q, :5] = (rb.addr("nbr_out"), rb.n_out, int(rb.K), pieces, buf.data_ptr() + off)
            rb._wgrad_plans[pieces] = buf[off:off + nbytes]
        _lib.check(lib.sv_wgrad_plan_build_batch(jobs.ctypes.data, len(todo), _lib.stream()), "sv_wgrad_plan_build_batch")
    for rb, cin, cout in items:
        rb.wgrad_plan(cin, cout)          # fills the per-layer cache from the per-table plans


def _dense_permutation(shape, strides):
    """True when (shape, strides) address every element of a dense buffer of prod(shape) elements exactly once."""
    expect = 1
    for sz, st in sorted(zip(shape, strides), key=lambda t: t[1]):
        if sz == 1:
            continue
        if st != expect:
            return False
        expect *= sz
    return True


def _conv_forward(features, weight_kio, rulebook, bn_partial=None):
    """-> (out (N_out, C_out), frag_bwd or None): the forward of one sparse convolution on contiguous fp32 features.  bn_partial: a callable that
    returns the device address for the BatchNorm partial sums; called (and the sums made) only when the planned kernel runs the layer."""
    w = weight_kio.detach()
    K, cin, cout = w.shape
    plan = rulebook.plan("fwd", cin, cout)
    if plan is not None:
        frag_fwd, frag_bwd = fragment_cache.get(weight_kio)     # not the detached copy: the cache keys on ._base
        return gather_gemm_planned(features, plan, frag_fwd, rulebook.n_out, K, cin, cout, bn_partial=bn_partial() if bn_partial else None), frag_bwd
    return gather_gemm(features, rulebook.nbr_out, w.permute(0, 2, 1), rulebook.n_out), None      # (K, C_out, C_in)


def _conv_backward(features, weight_kio, rb, frag_bwd, grad_out, need_input, need_weight):
    K, cin, cout = weight_kio.shape
    gf = gw = None
    if need_input:
        # dX[i] = sum_k dY[nbr_in[k][i]] @ W[k]^T  -> Wt[k][n=c_in][c=c_out] = W[k][c_in][c_out]: weight_kio itself
        plan = rb.plan("bwd", cout, cin)
        if plan is not None:
            fb = frag_bwd if frag_bwd is not None else fragment_cache.get(weight_kio)[1]
            gf = gather_gemm_planned(grad_out, plan, fb, rb.n_in, K, cout, cin)
        else:
            gf = gather_gemm(grad_out, rb.table_for_backward_data(), weight_kio.detach(), rb.n_in)
    if need_weight:
        gw = wgrad(features, rb.nbr_out, grad_out, K, cin, cout, like=weight_kio, plan=rb.wgrad_plan(cin, cout))
    return gf, gw


class SparseConvFunction(torch.autograd.Function):
    """features (N_in,C_in), weight_kio (K,C_in,C_out) -> (N_out,C_out). Backward: gather-GEMM over the input-major
    table for the data gradient and a deterministic row reduction for the weight gradient."""

    @staticmethod
    def forward(ctx, features, weight_kio, rulebook):
        _lib.require_cuda(features, weight_kio)
        features = features.contiguous().float()
        out, ctx.frag_bwd = _conv_forward(features, weight_kio, rulebook)
        ctx.rulebook = rulebook
        ctx.save_for_backward(features, weight_kio)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        features, weight_kio = ctx.saved_tensors
        gf, gw = _conv_backward(features, weight_kio, ctx.rulebook, ctx.frag_bwd, grad_out.contiguous().float(), ctx.needs_input_grad[0],
                                ctx.needs_input_grad[1])
        return gf, gw, None


class SparseConvBNReLUFunction(torch.autograd.Function):
    """SparseConvFunction followed by training-mode BatchNorm1d (+ ReLU) as ONE autograd node: the same five kernels per direction, half the
    Python / autograd bookkeeping per layer (the bench step is bound by the host thread that enqueues it).  Used by SparseSequential for a
    bias-free convolution directly followed by a fusable BatchNorm1d in training mode; the modules, their parameters and running statistics
    are the plain torch ones."""

    @staticmethod
    def forward(ctx, features, weight_kio, rulebook, gamma, beta, running_mean, running_var, momentum, eps, relu, num_batches_tracked):
        from . import norm
        _lib.require_cuda(features, weight_kio)
        features = features.contiguous().float()
        # the planned conv kernel leaves the BatchNorm partial sums of its output in the norm's scratch: no statistics pass over conv_out
        cout, used = weight_kio.shape[2], []
        conv_out, ctx.frag_bwd = _conv_forward(features, weight_kio, rulebook, bn_partial=(lambda: used.append(1) or norm.partial_address(cout, features.device))
                                               if norm.STATS_IN_CONV else None)
        y, mean, invstd = norm.bn_forward_raw(conv_out, gamma, beta, running_mean, running_var, momentum, eps, True, relu, num_batches_tracked,
                                              n_partials=_lib.load().sv_conv_planned_partials() if used else 0)
        ctx.rulebook, ctx.relu = rulebook, relu
        ctx.save_for_backward(features, weight_kio, conv_out, gamma, beta, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import norm
        features, weight_kio, conv_out, gamma, beta, mean, invstd = ctx.saved_tensors
        dconv, dgamma, dbeta = norm.bn_backward_raw(conv_out, dy.contiguous().float(), gamma, beta, mean, invstd, ctx.relu)
        gf, gw = _conv_backward(features, weight_kio, ctx.rulebook, ctx.frag_bwd, dconv, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gf, gw, None, (dgamma if gamma is not None else None), (dbeta if beta is not None else None), None, None, None, None, None, None


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


class DenseChannelsLastFunction(torch.autograd.Function):
    """dense().view(N, C * D, H, W) written in channels_last memory (sv_sparse_to_dense_nhwc): the (N, C D, H, W) tensor a channels_last 2-D backbone
    would otherwise make by copying the whole volume; the backward takes the gradient in that order too (any other layout is copied once)."""

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
        out = torch.empty((batch_size, h, w, c * d), dtype=torch.float32, device=dev)
        rc = lib.sv_sparse_to_dense_nhwc(_lib.ptr(features) if n else None, _lib.ptr(indices) if n else None, n, batch_size, c, d, h, w,
                                         _lib.ptr(scratch), _lib.ptr(out), _lib.stream())
        _lib.check(rc, "sv_sparse_to_dense_nhwc")
        ctx.save_for_backward(indices)
        ctx.dims = (batch_size, c, d, h, w)
        return out.permute(0, 3, 1, 2)                                   # (N, C D, H, W), channels_last strides

    @staticmethod
    def backward(ctx, grad):
        lib = _lib.load()
        (indices,) = ctx.saved_tensors
        b, c, d, h, w = ctx.dims
        n = indices.shape[0]
        grad = grad.float().permute(0, 2, 3, 1).contiguous()             # a view of a channels_last gradient
        out = torch.empty((n, c), dtype=torch.float32, device=grad.device)
        rc = lib.sv_dense_to_sparse_nhwc(_lib.ptr(grad), _lib.ptr(indices) if n else None, n, b, c, d, h, w, _lib.ptr(out) if n else None, _lib.stream())
        _lib.check(rc, "sv_dense_to_sparse_nhwc")
        return out, None, None, None


def sparse_to_dense_channels_last(features, indices, batch_size, spatial_shape):
    """(N, C * D, H, W) with channels_last strides, or None when the shape is not one sv_sparse_to_dense_nhwc takes (the caller then views dense())"""
    d, h, w = (int(s) for s in spatial_shape)
    if not features.is_cuda or not _lib.load().sv_sparse_to_dense_nhwc_applies(int(features.shape[1]), d, h, w):
        return None
    return DenseChannelsLastFunction.apply(features, indices, int(batch_size), [d, h, w])
