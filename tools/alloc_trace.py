#!/usr/bin/env python3
"""hipMalloc calls / reserved bytes of the caching allocator over the pipelined headline step (every 10 steps), and the sizes of the device allocations
made after the first 30 steps (torch.cuda.memory._record_memory_history) -- a steady-state step should make none."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
points, objects, scene, *_ = bench.make_inputs(0, dev)
model = bench.build_model(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, fused=True)
pre = bench.Prefetch(model, (points, objects, scene))
def st():
    m = torch.cuda.memory_stats()
    return m.get("num_device_alloc", 0), m.get("num_device_free", 0), m["reserved_bytes.all.current"] >> 20, m["allocated_bytes.all.current"] >> 20, m["allocated_bytes.all.peak"] >> 20
for i in range(int(os.environ.get("STEPS", "120"))):
    if i % 10 == 0 and os.environ.get("NO_SYNC") != "1":
        torch.cuda.synchronize(); print(i, "device allocs %d frees %d reserved %d MB allocated %d MB peak %d MB" % st(), flush=True)
    if i == 30:
        torch.cuda.memory._record_memory_history(max_entries=200000, stacks="python")
    bench.run_step_prefetched(model, opt, params, pre, 1)
torch.cuda.synchronize()
snap = torch.cuda.memory._snapshot()
ev = [t for tr in snap.get("device_traces", []) for t in tr if t.get("action") in ("segment_alloc", "segment_free")]
print(len(ev), "segment events after step 30")
from collections import Counter
c = Counter((e["action"], e["size"] >> 20, e.get("stream")) for e in ev)
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:20]:
    print(v, "x", k)
# who asked: the frames of the first few segment_alloc events
n = 0
for e in ev:
    if e["action"] == "segment_alloc" and n < 12:
        n += 1
        fr = [f'{os.path.basename(f["filename"])}:{f["line"]} {f["name"]}' for f in e.get("frames", []) if "site-packages" not in f["filename"] and "dist-packages" not in f["filename"]][:5]
        print(e["size"] >> 20, "MB on stream", e.get("stream"), "<-", " | ".join(fr))
