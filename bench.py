#!/usr/bin/env python3
"""bench.py — scenes/sec of the SEE-VCN hot path on MI355X (contract in the task description).

A step = one pass of the hot path over one batch per GPU: VCN_VC forward on 64 cropped objects (1024 pts each),
completed surfaces pasted into 16 KITTI-shaped ray-cast scenes (~17k pts each), dynamic voxelisation + mean VFE,
VoxelBackBone8x (sparse 3-D conv) + HeightCompression forward, loss, backward, SGD step.  Inputs are resident in
HBM before the timed region.  Scenes shard data-parallel (weak scaling); gradients are all-reduced over RCCL.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]
  python bench.py --config {stageA,second,pvrcnn,centerpoint} [--gpus N] [--steps K] [--warmup W]     side modes (bench_configs.py)

With --gpus N > 1 and no WORLD_SIZE in the environment the process is only a launcher: it starts N rank processes (one GPU each,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before anything touches the GPU and exits with their status -- the role of
`python -m torch.distributed.launch` in the reference's tools/scripts/dist_train.sh:17.  Under torchrun (WORLD_SIZE set) it is a rank.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

SCENES_PER_GPU = 16
OBJECTS_PER_GPU = 64
SCENE_N_AZ = 384                   # azimuth steps of the 64-beam ray cast: 20.9 k returns per scene inside the KITTI range
VCN_FLOP_PER_OBJECT = 1.976e9      # 987.8 M MAC / object, reference formulation
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
PEAK_HBM_GBS = 8000.0


def make_inputs(rank, device):
    import seevcn_amd.synth as synth
    pts, _ = synth.make_scene_batch(SCENES_PER_GPU, seed=2000 + 1000 * rank, n_az=SCENE_N_AZ)
    objs, _ = synth.make_object_batch(OBJECTS_PER_GPU, seed=1000 + 1000 * rank)
    # put the objects inside the KITTI range of their scene (x>0 half-plane) so their completed points voxelise
    objs = objs.copy()
    objs[:, :, 0] = np.abs(objs[:, :, 0])
    scene = (np.arange(OBJECTS_PER_GPU) // (OBJECTS_PER_GPU // SCENES_PER_GPU)).astype(np.float32)
    return (torch.from_numpy(pts).to(device), torch.from_numpy(objs).to(device), torch.from_numpy(scene).to(device), pts, objs, scene)


def build_model(device, seed=0):
    from seevcn_amd.pipeline import SceneStep
    from seevcn_amd.seeding import seeded_state_dict
    m = SceneStep()
    m.load_state_dict(seeded_state_dict(m, seed=seed))
    return m.to(device)


class _MeanSquare(torch.autograd.Function):
    """mean(x^2) over the dense BEV tensor: same value and gradient as x.square().mean().  The forward reads the tensor once and, while it does,
    writes the loss's own gradient (2 / n) x (sv_mean_square: one read + one write, fixed-order sums); the backward hands that tensor on, scaled by
    the upstream gradient on the device only when it is not 1 (sv_scale_by_device_scalar; in place, so ONE backward per forward).  The gradient tensor
    (577 MB here) lives from the forward to the backward -- the step's memory holds it beside the dense tensor, as a framework's saved activation would.  Round 4 ran a row-norm reduction forward (145 us for the
    577 MB tensor) and an elementwise product backward (210 us); SEEVCN_BENCH_LOSS=torch keeps that pair for A/B runs."""

    @staticmethod
    def forward(ctx, x):
        from seevcn_amd import _lib
        n = x.numel()
        if os.environ.get("SEEVCN_BENCH_LOSS") == "torch" or n % 4 or not x.is_contiguous():
            ctx.fused = False
            ctx.save_for_backward(x)
            flat = x.reshape(-1)
            if n % 1024 == 0:
                return torch.linalg.vector_norm(flat.view(-1, 1024), dim=1).square().sum() / n
            return torch.dot(flat, flat) / n
        lib = _lib.load()
        ctx.fused = True
        grad = torch.empty_like(x)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        scratch = _lib.workspace.scratch("bench_mean_square", lib.sv_mean_square_scratch_bytes(), x.device)
        _lib.check(lib.sv_mean_square(_lib.ptr(x), n, 1.0 / n, 2.0 / n, _lib.ptr(grad), _lib.ptr(out), _lib.ptr(scratch), _lib.stream()), "sv_mean_square")
        ctx.save_for_backward(grad)
        return out

    @staticmethod
    def backward(ctx, g):
        from seevcn_amd import _lib
        (t,) = ctx.saved_tensors
        if not ctx.fused:
            return t * (g * (2.0 / t.numel()))
        if getattr(ctx, "consumed", False):                        # the gradient tensor is scaled IN PLACE below: a second backward would scale it twice
            raise RuntimeError("bench._MeanSquare: backward ran twice on one forward (retain_graph is not supported by the fused stand-in loss)")
        ctx.consumed = True
        _lib.check(_lib.load().sv_scale_by_device_scalar(_lib.ptr(t), t.numel(), _lib.ptr(g.contiguous().float()), _lib.stream()), "sv_scale_by_device_scalar")
        return t


def loss_fn(bd):
    return _MeanSquare.apply(bd['spatial_features'])


def allreduce_grads(params, world):
    """The one exchange step of the data-parallel path: every gradient in ONE flat fp32 bucket (0.71 M values = 2.8 MB), one RCCL
    all-reduce over xGMI, averaged, copied back with one multi-tensor kernel (not one launch per parameter)."""
    if world == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat.div_(world)
    views, off = [], 0
    for g in grads:
        views.append(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
    torch._foreach_copy_(grads, views)


def run_step(model, opt, params, inputs, world):
    points, objects, scene = inputs
    opt.zero_grad(set_to_none=True)
    bd = model(points, objects, scene, SCENES_PER_GPU)
    loss = loss_fn(bd)
    loss.backward()
    allreduce_grads(params, world)
    opt.step()
    return loss


def device_spinup(seconds, device):
    """Untimed: keeps the GPU busy for `seconds` so that the measured steps run at sustained clocks (not a step, not counted as warm-up)."""
    if seconds <= 0:
        return
    a = torch.randn(4096, 4096, device=device)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            a = torch.nn.functional.normalize(a @ a, dim=1)
        torch.cuda.synchronize()


class Prefetch:
    """The input side of batch N + 1 (SceneStep.front: stage A, voxelisation, rulebooks, plans -- no weights involved) on a side stream while batch N
    trains on the main stream: the role of the reference's offline completion (SEE_VCN.py:85-115) and its dataloader workers
    (tools/train_utils/train_utils.py:22-29 fetches the next batch while the GPU works).  Every step still does one front and one compute.
    SEEVCN_BENCH_PREFETCH_DEPTH=2 (the default since round 5; 1 = one stage, A/B): two stages -- front_b(N + 1) (voxelise + index, the one blocking read)
    first, then front_a(N + 2) (stage A of the batch after next, no read), so that the read no longer waits for stage A's kernels.  Round 4 measured
    (tools/step_hosttime.py) 3.1 ms of host time per step instead of 4.1 and the same 4.13-4.21 ms step either way: the step was bound by the GPU
    (trained side alone 3.26 ms).  With round 5's shorter trained side (3.04 ms alone) the one-stage order's host thread (3.9 ms) is the longer chain and
    the two-stage order wins by 0.03-0.09 ms."""

    def __init__(self, model, inputs, threaded=False):
        from concurrent.futures import ThreadPoolExecutor
        self.model, self.inputs = model, inputs
        # side-stream priority: -1 (high) by default; SEEVCN_BENCH_SIDE_PRIORITY for A/B (HIP's range on this device is printed by tools/stream_priority.py)
        self.side = torch.cuda.Stream(priority=int(os.environ.get("SEEVCN_BENCH_SIDE_PRIORITY", "-1")))
        self.device = torch.cuda.current_device()
        self.pool = ThreadPoolExecutor(max_workers=1) if threaded else None
        if threaded:
            sys.setswitchinterval(float(os.environ.get("SEEVCN_BENCH_SWITCH_S", "2e-5")))   # default 5 ms: two launch-bound threads would take turns in 5 ms slices
        self.depth = int(os.environ.get("SEEVCN_BENCH_PREFETCH_DEPTH", "2"))          # round 5: 2 is the default (same box, three alternations: 4.06 / 4.01 / 3.99 at depth 1, 3.97 / 3.98 / 3.97 at depth 2)
        self.pending = None
        self.pasted = None                 # front_a's output waiting for its front_b (depth 2)
        self.first = True
        self._prime_allocator()

    def _prime_allocator(self):
        """Setup, not a step: spare blocks in the caching allocator's pools of both streams.  Buffers that cross from the input side's stream to the trained
        side's (the 450 MB index arena, voxel features, ...) go back to their pool only when the consumer's events have completed; until the pool holds a
        block or two more than the pipeline's depth a request finds none free and falls through to hipMalloc -- a 5-15 ms stall inside a step, tens of
        them while the pool fills (tools/alloc_trace.py).  A training run fills the pool in its first hundred steps; a 20-step measurement would spend
        them there.  SEEVCN_BENCH_PRIME=0: off (A/B)."""
        if os.environ.get("SEEVCN_BENCH_PRIME", "1") == "0":
            return
        sizes_mb = [512, 512, 256, 256, 128, 128, 64, 64, 64, 32, 32, 32, 16, 16, 16, 16]
        for stream in (self.side, torch.cuda.current_stream()):
            with torch.cuda.stream(stream):
                blocks = [torch.empty((mb << 20,), dtype=torch.uint8, device="cuda") for mb in sizes_mb]
                del blocks
        torch.cuda.synchronize()

    def _front(self):
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(self.side):
            if self.depth < 2:
                bd = self.model.front(*self.inputs, SCENES_PER_GPU)
            else:
                pts = self.pasted if self.pasted is not None else self.model.front_a(*self.inputs)
                bd = self.model.front_b(pts, SCENES_PER_GPU, flush=False)          # a check parked by the front_a below rides on the NEXT step's read
                ev = self.side.record_event()                                       # batch N + 1 is ready here: the main stream must not wait for stage A of N + 2
                self.pasted = self.model.front_a(*self.inputs)
                return bd, ev
            ev = self.side.record_event()
        return bd, ev

    def issue(self):
        if self.pending is not None and os.environ.get("SEEVCN_BENCH_REUSE_FRONT") == "1":
            return              # measurement aid ONLY (how long is the trained side alone?): the step is incomplete, its time is not a result
        if self.first:
            self.side.wait_stream(torch.cuda.current_stream())               # inputs / weights were produced on the main stream
            self.first = False
        self.pending = self.pool.submit(self._front) if self.pool is not None else self._front()

    def take(self):
        from seevcn_amd.pipeline import record_stream_tree
        if self.pending is None:
            self.issue()
        if os.environ.get("SEEVCN_BENCH_REUSE_FRONT") == "1" and isinstance(self.pending, tuple):
            return self.pending[0]
        bd, ev = self.pending.result() if self.pool is not None else self.pending
        if os.environ.get("SEEVCN_BENCH_REUSE_FRONT") == "1":
            self.pending = (bd, ev)
        main = torch.cuda.current_stream()
        main.wait_event(ev)
        record_stream_tree(bd, main)          # made on the side stream, read on the main one: the allocator must not hand the memory out early
        return bd

    def wait_issued(self):
        """Block until the worker thread has enqueued everything of the pending front (its GPU work is then covered by a device sync)."""
        if self.pool is not None and self.pending is not None and not isinstance(self.pending, tuple):
            self.pending.result()

    def close(self):
        from seevcn_amd import _lib
        if self.pool is not None:
            if self.pending is not None and not isinstance(self.pending, tuple):
                self.pending.result()
            self.pool.shutdown(wait=True)
        with torch.cuda.stream(self.side):
            _lib.flush_checks()                # the last front_a's parked check (its batch is never trained on)


def _compute_gen(model, opt, params, bd, world, out):
    """The trained side of one batch as a generator: yields between pieces so that the caller can enqueue them one at a time."""
    opt.zero_grad(set_to_none=True)
    res = yield from model.compute_stages(bd)
    loss = loss_fn(res)
    yield
    loss.backward()
    yield
    allreduce_grads(params, world)
    opt.step()
    out.append(loss)


def run_step_prefetched(model, opt, params, pre, world, interleave=True):
    """One step = compute(N) on the main stream + front(N + 1) on the side stream, enqueued by ONE host thread, whose time is what the step is
    made of (round 4: ~1.6 ms to enqueue the trained side, ~1.9 ms for the input side, against 3.8 ms of main-stream GPU time).  The input side has
    ONE blocking device -> host read (voxel count + the strided levels' site counts).  Order: forward + loss of batch N right away (the main
    stream has work from the start), then the input side of batch N + 1 up to its read; right BEFORE the read blocks -- sync hook of
    seevcn_amd._lib.host_int -- everything that is left of batch N (backward, exchange, optimiser) is enqueued, so the host never waits while the
    main stream could run dry; then the read, then the tables and plans of batch N + 1.  interleave False (A/B): the whole trained side first,
    then the input side."""
    from seevcn_amd import _lib
    # Back-pressure: the host reads only the input side's stream, so nothing ties it to the TRAINED side's -- when that stream is the slower one
    # the host and the input side run ahead of it by a step every dozen steps, each with a live 450 MB index arena and its voxel tensors (round 5:
    # 20 extra arenas = 9 GB hipMalloc'ed over 200 steps, tools/alloc_trace.py).  A step therefore first waits for the trained side of the step
    # before last (an event that has usually long completed): at most two trained steps are ever queued.
    inflight = pre.__dict__.setdefault("_done_events", [])
    if len(inflight) >= int(os.environ.get("SEEVCN_BENCH_MAX_INFLIGHT", "2")):
        t_w = time.perf_counter()
        inflight.pop(0).synchronize()
        pre.__dict__["_backpressure_s"] = pre.__dict__.get("_backpressure_s", 0.0) + (time.perf_counter() - t_w)      # host time blocked here: > 0 = the GPU is the slower side
    bd = pre.take()
    out = []
    gen = _compute_gen(model, opt, params, bd, world, out)
    main = torch.cuda.current_stream()

    def rest():
        # the hook is off while the pieces run: a device -> host read inside the trained side (a rulebook built lazily) must not re-enter
        mine = _lib.set_sync_hook(None)
        try:
            with torch.cuda.stream(main), torch.enable_grad():    # the hook runs inside the front's side-stream / no_grad context
                for _ in gen:
                    pass
        finally:
            _lib.set_sync_hook(mine)

    next(gen, None)                                           # forward + loss right away
    if not interleave:
        for _ in gen:
            pass
    prev = _lib.set_sync_hook(rest if interleave else None)
    try:
        pre.issue()
    finally:
        _lib.set_sync_hook(prev)
    for _ in gen:                                             # a front without a read (nothing to do) leaves the pieces here
        pass
    inflight.append(main.record_event())
    return out[0]


def measure_dominant_kernel(model, inputs, reps=5):
    """Live HIP-event timing of the dominant kernel (the fp32-MFMA GEMM of the VCN layers): every
    sv_gemm_bias_act launch of one VCN forward is bracketed by events on the launch stream."""
    from seevcn_amd.vcn.models import layers as L
    records = []
    orig = L.gemm

    def timed(a, w, *args, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = orig(a, w, *args, **kw)
        e.record()
        # rows actually computed: with a device-side row count (m_dev, the lazy-row forward) a.shape[0] is only the capacity
        records.append((kw.get("m_dev"), a.shape[0], 2.0 * a.shape[1] * w.shape[0], s, e))
        return out

    L.gemm = timed
    try:
        for _ in range(reps):
            model.vcn({'input': inputs[1]})
        torch.cuda.synchronize()
    finally:
        L.gemm = orig
    ms = sum(s.elapsed_time(e) for *_, s, e in records)
    executed = sum((int(m_dev.item()) if m_dev is not None else rows) * f for m_dev, rows, f, _, _ in records)
    launches = len(records)
    return ms / launches, executed / reps, launches // reps, ms / reps


def measure_spconv_kernel(model, opt, params, inputs, world, reps=3):
    """Live HIP-event timing of every planned sparse-conv launch (forward and data gradient) of `reps` extra, untimed steps ON THE LAUNCH-LIST PATH the
    timed steps run: the step's own forward / backward lists go through sv_run_ops_timed (events around every operation on the list's stream), so the
    launches timed are the instances the step really uses (input-transform instances in the forward, plain ones in the data gradient).  Grouped by
    kernel instance k_spconv_rs3<NT, KQ, G> (NT = min(C_out, 64) / 16 column tiles per wave, KQ = C_in / 16).  Returns the instance group with the
    largest total time: (name, avg launch ms, algorithmic flop per launch = 2*pairs*Cin*Cout averaged over its launches, algorithmic bytes per launch,
    launches per step, ms per step, extra) -- extra: the same four numbers for the equal-pieces weight gradient of the 64 -> 64 layers ("wgrad", or None) and
    the 2 * pairs * C_in * C_out of every planned conv / data gradient / weight gradient of a step ("planned_flop_per_step").  Pairs are counted from the
    rulebook table each launch used (outside the timed events)."""
    import ctypes
    from seevcn_amd import _lib
    from seevcn_amd.spconv import chain
    lists, dicts = [], []
    orig_run, orig_chain = chain._run, chain.run_chain

    def timed_run(rows, what):
        arr = np.array(rows, dtype=np.int64)
        ms = (ctypes.c_float * len(rows))()
        _lib.check(_lib.load().sv_run_ops_timed(arr.ctypes.data, len(rows), _lib.stream(), ctypes.cast(ms, ctypes.c_void_p)), what)
        lists.append((arr, list(ms)))

    def spy_chain(blocks, x):
        dicts.append(x.indice_dict)
        return orig_chain(blocks, x)

    chain._run, chain.run_chain = timed_run, spy_chain
    stream_was, chain.WGRAD_STREAM = chain.WGRAD_STREAM, False      # the events bracket single launches: these extra steps keep the weight gradients on the list's stream
    try:
        for _ in range(reps):
            run_step(model, opt, params, inputs, world)
        torch.cuda.synchronize()
    finally:
        chain._run, chain.run_chain, chain.WGRAD_STREAM = orig_run, orig_chain, stream_was
    if not lists:
        raise RuntimeError("the step did not run through the launch-list chain: nothing to time")
    # table address -> pairs of its rulebook (both directions of a layer share the pairs)
    pairs_of = {}
    for d in dicts:
        for rb in d.values():
            pairs = None
            for direction in ("fwd", "bwd"):
                for tp in ([rb._plans.get("fwd")] if direction == "fwd" or rb.subm else [rb._plans.get("bwd")]):
                    if tp is not None and tp.addr("rows") not in pairs_of:
                        if pairs is None:
                            pairs = int((rb.nbr_out >= 0).sum().item())
                        pairs_of[tp.addr("rows")] = pairs
    groups = {}
    nbr_pairs = {}
    for d in dicts:
        for rb in d.values():
            nbr_pairs[rb.addr("nbr_out")] = rb
    wg = [0.0, 0.0, 0.0, 0]          # the equal-pieces weight gradient of the 64 -> 64 layers (k_spconv_wgrad_eq<4, 4>): ms, flop, SURVEY 8(d) bytes, launches
    all_flop = 0.0                   # every conv / data-gradient / weight-gradient operation of the lists: 2 * pairs * C_in * C_out each
    for arr, ms in lists:
        for row, t in zip(arr, ms):
            if int(row[0]) in (chain.OP_WGRAD, chain.OP_WGRAD_DEFERRED):
                K, cin, cout, n_in, n_out = int(row[1]), int(row[2]), int(row[3]), int(row[4]), int(row[9])
                rb = nbr_pairs.get(int(row[18]))
                if rb is None:
                    continue
                tp = rb._plans.get("fwd")
                key = tp.addr("rows") if tp is not None else None
                if key in pairs_of:
                    pairs = pairs_of[key]
                else:
                    pairs = int((rb.nbr_out >= 0).sum().item())
                all_flop += 2.0 * pairs * cin * cout
                if int(row[22]) and cin == 64 and cout == 64:                              # p[5]: an equal-pieces plan -> k_spconv_wgrad_eq<4, 4>
                    wg[0] += t
                    wg[1] += 2.0 * pairs * cin * cout
                    wg[2] += 4.0 * (n_in * cin + n_out * cout) + 4.0 * K * cin * cout + 8.0 * pairs
                    wg[3] += 1
                continue
            if int(row[0]) == chain.OP_CONV_PLAIN:
                continue
            if int(row[0]) not in (chain.OP_CONV_PLANNED, chain.OP_DGRAD_PLANNED_BN):     # the same kernel; the second form's epilogue also makes a BatchNorm's backward sums
                continue
            K, kd, nc = int(row[2]), int(row[3]), int(row[4])
            n_src, n_rows, table = int(row[9]), int(row[10]), int(row[18])
            pairs = pairs_of[table]
            g = groups.setdefault((min(nc, 64) // 16, kd // 16), [0.0, 0.0, 0.0, 0])
            g[0] += t
            g[1] += 2.0 * pairs * kd * nc
            g[2] += 4.0 * (n_src * kd + n_rows * nc) + 4.0 * K * kd * nc + 8.0 * pairs    # features in+out, weights, rulebook pairs
            g[3] += 1
            all_flop += 2.0 * pairs * kd * nc
    (nt, kq), (ms, flop, byt, n) = max(groups.items(), key=lambda kv: kv[1][0])
    extra = {"wgrad": (wg[0] / max(wg[3], 1), wg[1] / max(wg[3], 1), wg[2] / max(wg[3], 1), wg[3] // reps) if wg[3] else None,
             "planned_flop_per_step": all_flop / reps}
    return f"k_spconv_rs3<{nt}, {kq}, ", ms / n, flop / n, byt / n, n // reps, ms / reps, extra


def cpu_baseline(pts_np, objs_np, scene_np, n_objects=4, n_scenes=1, warmup=3, timed=10):
    """Oracle (CPU port) on a bounded sample of the same workload: VCN + post-processing on n_objects objects and one scene's
    merge -> voxelise -> VoxelBackBone8x forward/backward (numpy sparse conv inside torch-CPU autograd for BN/ReLU).
    SURVEY 8(d): warm-up passes, then the median of the timed passes (each pass = the whole sample)."""
    from oracle import vcn as ovcn, voxelize as ovox, postprocess as opp, spconv_train as ost
    from seevcn_amd.pipeline import KITTI
    from seevcn_amd.pcdet.models import backbones_3d
    from seevcn_amd.seeding import seeded_state_dict
    import seevcn_amd.vcn as V
    threads = torch.get_num_threads()
    vsd = seeded_state_dict(V.MODELS.build({'NAME': 'VCN_VC'}), seed=0)
    g = KITTI
    m = backbones_3d.__all__['VoxelBackBone8x']({}, 3, g['grid_size'])
    sd = {k: v.numpy() for k, v in seeded_state_dict(m, seed=0).items()}
    per_scene_objs = OBJECTS_PER_GPU // SCENES_PER_GPU

    def one_pass():
        t0 = time.perf_counter()
        coarse = ovcn.vcn_vc_forward(vsd, torch.from_numpy(objs_np[:n_objects]))['coarse'].numpy()
        surface = opp.get_partial_mesh_batch(objs_np[:n_objects], coarse, k=30)
        coarse = opp.get_largest_cluster_batch(surface, eps=0.4, min_points=2).astype(np.float32)      # 'clustered'
        t_vcn = (time.perf_counter() - t0) / n_objects            # s / object
        t1 = time.perf_counter()
        sel = pts_np[pts_np[:, 0] < n_scenes]
        paste = np.concatenate([np.repeat(scene_np[:n_objects, None, None], 1024, 1), coarse], axis=2).reshape(-1, 4)
        paste = paste[paste[:, 0] < n_scenes]
        if len(paste):
            inst = np.unique(paste.astype(np.float32), axis=0)
            near = np.zeros(len(sel), bool)
            for b in np.unique(inst[:, 0]):
                qm = sel[:, 0] == b
                near[qm] = opp.replace_with_completed_pts(sel[qm, 1:4], inst[inst[:, 0] == b, 1:4], 0.1)[1]
            allp = np.concatenate([inst, sel[~near]], 0)
        else:
            allp = sel
        feats, coords, _ = ovox.dynamic_mean_vfe(allp, g['point_cloud_range'], g['voxel_size'], g['grid_size'])
        dense, _, _ = ost.backbone8x_train_chain(sd, feats, coords, n_scenes, m.sparse_shape, dtype=torch.float32)
        dense.square().mean().backward()
        t_scene = (time.perf_counter() - t1) / n_scenes
        return t_vcn, t_scene

    for _ in range(warmup):
        one_pass()
    runs = [one_pass() for _ in range(timed)]
    t_vcn = float(np.median([r[0] for r in runs]))
    t_scene = float(np.median([r[1] for r in runs]))
    sec_per_scene = t_scene + per_scene_objs * t_vcn
    return {"value": round(1.0 / sec_per_scene, 4), "unit": "scenes/sec", "cores": threads, "kind": "port",
            "sample": f"oracle (numpy/torch-CPU, {threads} threads), {warmup} warm-up + {timed} timed passes, medians: VCN_VC fwd + surface select + "
                      f"DBSCAN on {n_objects} objects ({t_vcn:.3f} s/object) + {n_scenes} scene merge+voxelise+VoxelBackBone8x fwd+bwd "
                      f"({t_scene:.2f} s/scene); scaled to {per_scene_objs} objects/scene"}


def rank_identity(device):
    """What this rank runs on, for the record of an N > 1 run: (hostname, visible device index, device UUID, PCI bus id)."""
    if device is None or device.type != "cuda":
        return (socket.gethostname(), -1, f"cpu-{os.getpid()}", "")
    pr = torch.cuda.get_device_properties(device)
    return (socket.gethostname(), int(device.index or 0), str(getattr(pr, "uuid", "")), f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}")


def gather_rank_identities(device, world):
    """All-gathered over the process group: {'world_size_seen': ranks that answered, 'devices': sorted per-rank identities, 'distinct_devices': n}.
    A SCALE record can then PROVE that N ranks ran on N different GPUs (or, in the shared-GPU test mode, that they did not)."""
    me = rank_identity(device)
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, me)
    else:
        got = [me]
    devs = sorted(f"{h}/{i}/{u}/{b}" for h, i, u, b in got)
    return {"world_size_seen": len(got), "devices": devs, "distinct_devices": len(set(devs))}


def _free_port():
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (rank env set, one GPU each) and relay their
    exit status; rank 0 prints the JSON line on the inherited stdout.  The parent never calls into HIP (a process that has initialised
    the GPU must not be replaced or forked), it only waits."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        for p in procs:
            rc = max(rc, abs(p.wait()))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def dry_run_rank(args, rank, world):
    """The N > 1 control flow of a rank without a GPU (CPU tests): process group over gloo, the flat-bucket exchange on fake gradients,
    the barrier / MAX-over-ranks timing protocol, rank 0's JSON line, barrier, teardown."""
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    params = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        allreduce_grads(params, world)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    want = sum(range(1, world + 1)) / world
    ok = all(torch.allclose(p.grad, torch.full_like(p, want * (i + 1))) for i, p in enumerate(params))
    ident = gather_rank_identities(None, world)
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "exchange_ok": bool(ok),
                          "elapsed_s": round(elapsed, 4), "ranks": ident}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # 0.8 s timed: 20-step runs of this 4 ms step scattered by +-5 % on one box
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scenes-per-gpu", type=int, default=None, help="override the 16 scenes per GPU (tests)")
    ap.add_argument("--objects-per-gpu", type=int, default=None, help="override the 64 objects per GPU (tests)")
    ap.add_argument("--no-side-modes", action="store_true", help="skip the brief runs of the other BASELINE configs after the headline")
    ap.add_argument("--no-kernel-rooflines", action="store_true", help="skip the per-kernel event timing after the timed region (tests)")
    ap.add_argument("--dry-run", action="store_true", help="CPU-only control flow of the multi-rank path (gloo), no kernels")
    ap.add_argument("--no-prefetch", action="store_true", help="run the input side (stage A, voxelise, rulebooks) in line on the main stream")
    ap.add_argument("--no-interleave", action="store_true", help="enqueue the WHOLE trained side before the input side instead of its backward half inside the input side's read (A/B)")
    ap.add_argument("--prefetch-thread", action="store_true", help="input side issued by a host thread of its own (A/B: measured slower, "
                    "two launch-bound Python threads take turns on the GIL)")
    ap.add_argument("--spinup", type=float, default=0.0, help="seconds of untimed device spin-up (dense GEMMs) before the warm-up steps (A/B switch: a GPU "
                    "that idled may start below its sustained clocks; measured here: no effect, default off)")
    ap.add_argument("--roofline-only", type=int, default=0, metavar="REPS", help="no timed steps: warm up, then REPS passes of the launch-list timing the "
                    "`roofline` block is made of (one stream, weight gradients in line) and print that block -- the command tools/evidence.sh puts under "
                    "rocprofv3 --kernel-trace --stats, so that profiles/ holds the trace of exactly the launches the line's roofline times")
    ap.add_argument("--config", default="main", choices=("main", "stageA", "second", "pvrcnn", "centerpoint"),
                    help="main (default): the headline step; the others are side modes over the other BASELINE configs (bench_configs.py)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    global SCENES_PER_GPU, OBJECTS_PER_GPU
    if args.scenes_per_gpu:
        SCENES_PER_GPU = args.scenes_per_gpu
    if args.objects_per_gpu:
        OBJECTS_PER_GPU = args.objects_per_gpu
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dry_run:
        sys.exit(dry_run_rank(args, rank, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # SEEVCN_BENCH_SHARE_GPU=1 + SEEVCN_BENCH_BACKEND=gloo: every rank on cuda:0 (RCCL refuses two ranks per device) -- lets a 1-GPU box
    # run the whole N > 1 control flow of this file end to end (tests/test_dist.py)
    if os.environ.get("SEEVCN_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("SEEVCN_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.config != "main":
        side_mode(args, rank, world, device)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    points, objects, scene, pts_np, objs_np, scene_np = make_inputs(rank, device)
    model = build_model(device).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, fused=True)     # one multi-tensor kernel instead of ~70 small launches
    inputs = (points, objects, scene)

    if args.roofline_only:
        for _ in range(max(args.warmup, 3)):
            run_step(model, opt, params, inputs, world)
        torch.cuda.synchronize()
        name, c_ms, c_flop, c_bytes, c_launches, c_step_ms, _ = measure_spconv_kernel(model, opt, params, inputs, 1, reps=args.roofline_only)
        if rank == 0:
            c_ach = c_flop / (c_ms * 1e-3) / 1e12
            print(json.dumps({"roofline_only": True, "reps": args.roofline_only, "kernel": f"{name}G>", "avg_launch_ms": round(c_ms, 4), "launches_per_step": c_launches,
                              "achieved": round(c_ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "frac": round(c_ach / PEAK_F32_MFMA_TFLOPS, 4),
                              "algorithmic_flop_per_launch": round(c_flop)}))
        return

    device_spinup(args.spinup, device)
    # SEEVCN_BENCH_MAIN_PRIORITY (A/B): the trained side on a stream of its own with this priority instead of the default stream (priority 0)
    import contextlib
    main_ctx = contextlib.nullcontext()
    if os.environ.get("SEEVCN_BENCH_MAIN_PRIORITY") is not None:
        main_stream = torch.cuda.Stream(priority=int(os.environ["SEEVCN_BENCH_MAIN_PRIORITY"]))
        main_stream.wait_stream(torch.cuda.current_stream())
        main_ctx = torch.cuda.stream(main_stream)
    main_ctx.__enter__()
    pre = None if args.no_prefetch else Prefetch(model, inputs, threaded=args.prefetch_thread)
    step = (lambda: run_step(model, opt, params, inputs, world)) if pre is None else (lambda: run_step_prefetched(model, opt, params, pre, world, not args.no_interleave))
    for _ in range(args.warmup):
        step()
    if pre is not None:
        pre.wait_issued()              # the front of the first timed batch: enqueued and (next line) finished before the clock starts
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    import gc
    gc0 = [dict(g) for g in gc.get_stats()]
    dev_allocs0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    if pre is not None:
        pre.__dict__["_backpressure_s"] = 0.0
    t0 = time.perf_counter()
    stamps = []                        # host clock after every step's enqueue (a pipelined step ends in a blocking read: stamps follow the steps): one
    for _ in range(args.steps):        # perf_counter call per step, reported as the slowest / median step interval so that a transient shows in the line
        step()
        stamps.append(time.perf_counter())
    if pre is not None:
        pre.wait_issued()              # K steps = K computes + K fronts: the last front issued belongs to the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    backpressure_ms = pre.__dict__.get("_backpressure_s", 0.0) / args.steps * 1e3 if pre is not None else None
    if pre is not None:
        pre.close()
    main_ctx.__exit__(None, None, None)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ident = gather_rank_identities(device, world)            # collective: every rank, before rank 0 goes off measuring alone
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        gaps = np.diff(np.array([t0] + stamps)) * 1e3
        dev_allocs = torch.cuda.memory_stats().get("num_device_alloc", 0) - dev_allocs0
        gc1 = gc.get_stats()
        gc_runs = [b["collections"] - a["collections"] for a, b in zip(gc0, gc1)]
        gc_freed = sum(b["collected"] - a["collected"] for a, b in zip(gc0, gc1))
        scenes = SCENES_PER_GPU * world * args.steps
        in_range = int(((pts_np[:, 1] >= 0) & (pts_np[:, 1] < 70.4) & (np.abs(pts_np[:, 2]) < 40) & (pts_np[:, 3] >= -3) & (pts_np[:, 3] < 1)).sum())
        out = {
            "metric": "scenes/sec (VCN+voxel+spconv fwd+bwd)", "value": round(scenes / elapsed, 3), "unit": "scenes/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "VCN_VC fwd on 64 objects x 1024 pts -> kNN surface select (k=30) -> largest DBSCAN cluster (eps 0.4) -> unique + "
                                   f"replace (<0.1 m) into 16 KITTI-shaped 64-beam scenes ({in_range / SCENES_PER_GPU / 1e3:.1f}k pts in range each) -> "
                                   "DynMeanVFE -> VoxelBackBone8x + HeightCompression fwd+bwd + SGD (BASELINE configs[1]+[2])",
                       "scenes_per_gpu": SCENES_PER_GPU, "objects_per_gpu": OBJECTS_PER_GPU, "points_per_object": 1024,
                       "points_in_range_per_scene": round(in_range / SCENES_PER_GPU),
                       "geometry": "KITTI [0,-40,-3,70.4,40,1] @ [0.05,0.05,0.1] -> sparse [41,1600,1408]", "parallelism": f"dp{world}",
                       "input_side": "in line" if pre is None else "side stream, one batch ahead (every step = 1 front + 1 compute)"},
            "completed_objects_per_sec": round(OBJECTS_PER_GPU * world * args.steps / elapsed, 1), "device_spinup_s": args.spinup,
            "step_interval_ms": {"median": round(float(np.median(gaps)), 3), "max": round(float(gaps.max()), 3), "first_10": [round(float(g), 2) for g in gaps[:10]],
                                 "slow_steps": int((gaps > 2 * np.median(gaps)).sum()),
                                 "python_gc_runs_by_generation": gc_runs, "python_gc_objects_freed": gc_freed, "hipMalloc_calls_in_timed_steps": int(dev_allocs),
                                 # host time per step spent BLOCKED on the trained side of the step before last (run_step_prefetched's back-pressure): the part of
                                 # the step the one host thread has to spare -- ~0 would mean the host's enqueue time, not the GPU, sets the step
                                 "host_blocked_on_gpu_ms": None if backpressure_ms is None else round(backpressure_ms, 3)},
            "ranks": ident, "backend": (backend if world > 1 else None),
        }
        if not args.no_kernel_rooflines:
            # rank 0 only, so NO collective inside: the measured steps run with world = 1 (the exchange step is skipped)
            kernel_rooflines(out, model, opt, params, inputs)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pts_np, objs_np, scene_np)
        if not args.no_side_modes and world == 1:
            # the other BASELINE configs, a few steps each, AFTER the headline was measured (its numbers above are final): same protocol as
            # `--config X`, so that the driver's one command sees them too
            del model, opt, params, inputs, points, objects, scene, pre, step
            out["side_modes"] = side_modes_brief(device)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()             # the other ranks wait here while rank 0 measures and prints
        dist.destroy_process_group()


def side_modes_brief(device, warmup=6, steps=15):
    """{config: ms_per_step, ...} of bench_configs' detector train steps (second = BASELINE configs[2], pvrcnn = [3], centerpoint = [4]) and
    stage A alone ([1]); N = 1 only, warm-up includes MIOpen's solver search."""
    import gc
    import bench_configs
    sys.modules.setdefault("bench", sys.modules[__name__])
    res = {}
    for name in ("stageA", "second", "pvrcnn", "centerpoint"):
        gc.collect()
        torch.cuda.empty_cache()
        try:
            step, units, unit, metric, config = bench_configs.build(name, 0, device)
            w, k = (10, 20) if name == "stageA" else (warmup, steps)       # a 1 ms inference step: more of them cost nothing and cover the allocator's settling
            for _ in range(w):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / k * 1e3
            res[name] = {"ms_per_step": round(ms, 3), "value": round(units / (ms * 1e-3), 2), "unit": unit, "steps": k, "warmup": w,
                         "workload": config["workload"]}
            del step
        except Exception as e:                                     # a side mode must never take the headline line down with it
            res[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return res


def side_mode(args, rank, world, device):
    """--config X: the same timing protocol over one of the other BASELINE configs (bench_configs.py); `value` = units of all ranks / max time."""
    import bench_configs
    sys.modules.setdefault("bench", sys.modules[__name__])                 # bench_configs reaches allreduce_grads / SCENE_N_AZ through `import bench`
    step, units, unit, metric, config = bench_configs.build(args.config, rank, device, scenes=args.scenes_per_gpu)
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ident = gather_rank_identities(device, world)
    if rank == 0:
        config["parallelism"] = f"dp{world}"
        config["ranks"] = ident
        print(json.dumps({"metric": metric, "value": round(units * world * args.steps / elapsed, 3), "unit": unit, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": config, "side_mode": args.config}), flush=True)


def kernel_rooflines(out, model, opt, params, inputs):
    avg_ms, executed_flop, launches, vcn_gemm_ms = measure_dominant_kernel(model, inputs)
    algo_flop_per_launch = VCN_FLOP_PER_OBJECT * OBJECTS_PER_GPU / launches
    achieved = algo_flop_per_launch / (avg_ms * 1e-3) / 1e12
    exec_tf = executed_flop / (vcn_gemm_ms * 1e-3) / 1e12
    if exec_tf > PEAK_F32_MFMA_TFLOPS:
        raise RuntimeError(f"VCN GEMM accounting: {exec_tf:.1f} TFLOP/s executed is above the fp32 MFMA peak -- rows or time are miscounted")
    vcn_roof = {"bound": "mfma", "kernel": "k_gemm_f32 (sv_gemm_bias_act[_ragged], v_mfma_f32_32x32x2_f32)",
                "achieved": round(exec_tf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(exec_tf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                "launches_per_step": launches, "avg_launch_ms": round(avg_ms, 4), "ms_per_step": round(vcn_gemm_ms, 3),
                "note": "achieved = EXECUTED flops (matrix-core utilisation). Each object's 1024 rows are ResamplePoints copies of "
                        "30-400 points: the per-point layers run on the distinct rows only (bit-identical output), so the time per "
                        "object is far below what SURVEY 8(d)'s 1.976 GFLOP/object implies; the algorithmic figure below is NOT a rate",
                "algorithmic_flop_per_launch": algo_flop_per_launch, "algorithmic_tflops": round(achieved, 2)}
    # dominant kernel by time in the step: the register-stationary sparse-conv gather-GEMM (forward + data gradient)
    name, c_ms, c_flop, c_bytes, c_launches, c_step_ms, extra = measure_spconv_kernel(model, opt, params, inputs, 1)
    c_ach = c_flop / (c_ms * 1e-3) / 1e12
    out["roofline"] = {"bound": "mfma", "kernel": f"{name}G> (sv_sparse_conv_gather_gemm_planned, v_mfma_f32_16x16x4_f32; G = tiles per wave, 1-4 by layer size)",
                       "achieved": round(c_ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(c_ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                       "launches_per_step": c_launches, "avg_launch_ms": round(c_ms, 4), "ms_per_step": round(c_step_ms, 3),
                       "algorithmic_flop_per_launch": round(c_flop), "algorithmic_bytes_per_launch": round(c_bytes),
                       "algorithmic_GBps": round(c_bytes / (c_ms * 1e-3) / 1e9, 1)}
    if extra.get("wgrad"):
        w_ms, w_flop, w_bytes, w_n = extra["wgrad"]
        w_ach = w_flop / (w_ms * 1e-3) / 1e12
        out["roofline_wgrad"] = {"bound": "mfma", "kernel": "k_spconv_wgrad_eq<4, 4> (sv_sparse_conv_wgrad_planned_stage1, v_mfma_f32_16x16x4_f32; 64 -> 64 layers)",
                                 "achieved": round(w_ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(w_ach / PEAK_F32_MFMA_TFLOPS, 4),
                                 "traffic": None, "launches_per_step": w_n, "avg_launch_ms": round(w_ms, 4),
                                 "algorithmic_flop_per_launch": round(w_flop), "algorithmic_bytes_per_launch": round(w_bytes),
                                 "note": "event-timed on the list's own stream with the weight gradients in line (in the pipelined step they run on a second stream beside the data gradients)"}
    if extra.get("planned_flop_per_step"):
        # whole step against the matrix peak: every planned conv / data gradient / weight gradient's 2 * pairs * C_in * C_out over the driver-timed step
        fl = extra["planned_flop_per_step"]
        out["mfma_frac_step"] = {"sparse_conv_flop_per_step": round(fl), "ms_per_step": out["ms_per_step"],
                                 "achieved": round(fl / (out["ms_per_step"] * 1e-3) / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(fl / (out["ms_per_step"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
    # HBM traffic of that kernel: PMC counters cannot be read from inside the process; the newest committed PMC pass over this
    # same command (tools/traffic.sh -> profiles/*_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes, x2 on reads for gfx950)
    import glob
    for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))[-1:]:
        with open(tf) as fh:
            kernels = json.load(fh).get("kernels", {})
        hit = [v for k, v in kernels.items() if name in k]          # the G = 2 / 3 / 4 instances of this channel shape
        if hit:
            n_disp = sum(v["dispatches"] for v in hit)
            out["roofline"]["traffic"] = round(sum(v["hbm_bytes_per_dispatch"] * v["dispatches"] for v in hit) / n_disp)
            out["roofline"]["traffic_source"] = os.path.relpath(tf, ROOT)
        hitw = [v for k, v in kernels.items() if "k_spconv_wgrad_eq<4, 4" in k]
        if hitw and "roofline_wgrad" in out:
            n_disp = sum(v["dispatches"] for v in hitw)
            out["roofline_wgrad"]["traffic"] = round(sum(v["hbm_bytes_per_dispatch"] * v["dispatches"] for v in hitw) / n_disp)
            out["roofline_wgrad"]["traffic_source"] = os.path.relpath(tf, ROOT)
    out["roofline_vcn_gemm"] = vcn_roof


if __name__ == "__main__":
    main()
