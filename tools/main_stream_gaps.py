#!/usr/bin/env python3
"""Where the TRAINED side's stream spends a pipelined headline step, from device timestamps (HIP events on that stream, no profiler): forward + loss,
the wait for the host to enqueue the backward, backward + optimiser, the wait for the next step's forward.  The stream's own kernels take ~3.0 ms of a
3.7 ms step; this says which waits make up the rest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
points, objects, scene, *_ = bench.make_inputs(0, dev)
model = bench.build_model(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, fused=True)
pre = bench.Prefetch(model, (points, objects, scene))
marks = []
orig = bench._compute_gen
def gen(model_, opt_, params_, bd, world, out):
    m = {}
    ev = lambda: torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
    m["f0"] = ev()
    opt_.zero_grad(set_to_none=True)
    res = yield from model_.compute_stages(bd)
    loss = bench.loss_fn(res)
    m["f1"] = ev()
    yield
    m["b0"] = ev()
    loss.backward()
    yield
    bench.allreduce_grads(params_, world)
    opt_.step()
    m["b1"] = ev()
    out.append(loss)
    marks.append(m)
bench._compute_gen = gen
for _ in range(20): bench.run_step_prefetched(model, opt, params, pre, 1)
torch.cuda.synchronize(); marks.clear()
N = 100
for _ in range(N): bench.run_step_prefetched(model, opt, params, pre, 1)
torch.cuda.synchronize()
fwd = [m["f0"].elapsed_time(m["f1"]) for m in marks]
mid = [m["f1"].elapsed_time(m["b0"]) for m in marks]
bwd = [m["b0"].elapsed_time(m["b1"]) for m in marks]
nxt = [a["b1"].elapsed_time(b["f0"]) for a, b in zip(marks[:-1], marks[1:])]
tot = marks[0]["f0"].elapsed_time(marks[-1]["f0"]) / (N - 1)
print(f"step (stream clock)               {tot:6.3f} ms")
for name, v in (("forward + loss", fwd), ("loss end -> backward start", mid), ("backward + optimiser", bwd), ("optimiser end -> next forward start", nxt)):
    print(f"{name:34s}{np.median(v):6.3f} ms   (p10 {np.percentile(v, 10):.3f}, p90 {np.percentile(v, 90):.3f})")
