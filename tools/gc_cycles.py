#!/usr/bin/env python3
"""Reference cycles a pipelined headline step leaves behind (objects only the cyclic collector can free): 10 steps with the collector off, then one
collection with DEBUG_SAVEALL -- types of the garbage and the device tensors caught in it.  A tensor in a cycle stays allocated until a generation-1 / -2
collection happens to run: the caching allocator then grows (hipMalloc inside a step: a 5-15 ms stall) although the step's live set is constant."""
import gc, os, sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
points, objects, scene, *_ = bench.make_inputs(0, dev)
model = bench.build_model(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, fused=True)
pre = bench.Prefetch(model, (points, objects, scene))
for _ in range(10):
    bench.run_step_prefetched(model, opt, params, pre, 1)
torch.cuda.synchronize(); gc.collect(); gc.disable()
a0 = torch.cuda.memory_allocated()
for _ in range(10):
    bench.run_step_prefetched(model, opt, params, pre, 1)
torch.cuda.synchronize()
a1 = torch.cuda.memory_allocated()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
print(f"10 steps: {n} unreachable objects = {n / 10:.0f} per step; allocated {a0 >> 20} -> {a1 >> 20} MB with the collector off")
c = Counter(type(o).__module__ + "." + type(o).__qualname__ for o in gc.garbage)
for k, v in c.most_common(25):
    print(f"{v:6d}  {k}")
tens = [o for o in gc.garbage if isinstance(o, torch.Tensor) and o.is_cuda]
print(len(tens), "device tensors in cycles,", sum(t.untyped_storage().nbytes() for t in tens) >> 20, "MB (storages counted per tensor)")
for t in sorted(tens, key=lambda t: -t.untyped_storage().nbytes())[:8]:
    print("   ", tuple(t.shape), t.dtype, t.untyped_storage().nbytes() >> 20, "MB")
