"""Where does a side-mode step synchronise the host with the GPU?  torch's sync debug mode warns at every synchronising call; the warnings of
ONE steady step are grouped by the innermost frame inside this repository.   python tools/sync_trace.py {main,second,pvrcnn,centerpoint}"""
import collections
import os
import sys
import traceback
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bench_configs  # noqa: E402

sys.modules.setdefault("bench", bench)
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
which = sys.argv[1] if len(sys.argv) > 1 else "pvrcnn"
if which == "main":
    # the headline step as bench.py runs it: trained side of batch N on the main stream, input side of batch N + 1 on the side stream
    points, objects, scene, *_ = bench.make_inputs(0, device)
    model = bench.build_model(device).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, fused=True)
    pre = bench.Prefetch(model, (points, objects, scene))

    def step():
        return bench.run_step_prefetched(model, opt, params, pre, 1)
else:
    step, *_ = bench_configs.build(which, 0, device)
for _ in range(3):
    step()
torch.cuda.synchronize()
hits = collections.Counter()


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    frames = [f for f in traceback.extract_stack()[:-1] if f.filename.startswith(ROOT) and "sync_trace" not in f.filename]
    key = " <- ".join(f"{os.path.relpath(f.filename, ROOT)}:{f.lineno}" for f in frames[-3:][::-1])
    hits[key] += 1


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
step()
torch.cuda.set_sync_debug_mode(0)
torch.cuda.synchronize()
print("synchronising calls in one step:", sum(hits.values()))
for k, v in hits.most_common(60):
    print(f"{v:4d}  {k}")
