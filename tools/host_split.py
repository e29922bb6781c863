import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from seevcn_amd.spconv import chain
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
points, objects, scene, *_ = bench.make_inputs(0, dev)
model = bench.build_model(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, fused=True)
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try: return f(*a, **k)
        finally: T.setdefault(key, []).append(time.perf_counter() - t0)
    setattr(obj, name, g)
wrap(chain, "_run", "C: sv_run_ops (fwd list / bwd list when one stream)")
wrap(chain, "_run_two_streams", "C: sv_run_ops_two_streams (bwd list)")
fwd0, bwd0 = chain.SparseChainFunction.forward, chain.SparseChainFunction.backward
def fwd(ctx, *a):
    t0 = time.perf_counter()
    try: return fwd0(ctx, *a)
    finally: T.setdefault("py+C: SparseChainFunction.forward", []).append(time.perf_counter() - t0)
def bwd(ctx, *g):
    t0 = time.perf_counter()
    try: return bwd0(ctx, *g)
    finally: T.setdefault("py+C: SparseChainFunction.backward", []).append(time.perf_counter() - t0)
chain.SparseChainFunction.forward = staticmethod(fwd); chain.SparseChainFunction.backward = staticmethod(bwd)
wrap(model, "compute", "SceneStep.compute (backbone fwd + dense)")
wrap(model, "front_a", "front_a")
wrap(model, "front_b", "front_b")
wrap(opt, "step", "opt.step")
pre = bench.Prefetch(model, (points, objects, scene))
for _ in range(10): bench.run_step_prefetched(model, opt, params, pre, 1)
torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter()
for _ in range(50): bench.run_step_prefetched(model, opt, params, pre, 1)
torch.cuda.synchronize()
print("wall", (time.perf_counter()-t0)/50*1e3)
for k, v in T.items(): print(f"{k:60s} {np.median(v)*1e3:7.3f} ms x{len(v)/50:.0f}")
