#!/usr/bin/env python3
"""Which source lines of this repository issue the (forward) device work of a side-mode step?  One steady step under a TorchDispatchMode: every
dispatched aten op that is not a pure view is attributed to the innermost Python frame inside the repository.  (Backward ops run from the
autograd engine and have no Python frame: their number mirrors the differentiable forward ops.)   python tools/launch_sites.py pvrcnn [top]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bench_configs  # noqa: E402

sys.modules.setdefault("bench", bench)
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
step, *_ = bench_configs.build(sys.argv[1] if len(sys.argv) > 1 else "pvrcnn", 0, device)
for _ in range(3):
    step()
torch.cuda.synchronize()
VIEWS = {"view", "_unsafe_view", "reshape", "slice", "select", "as_strided", "detach", "alias", "expand", "permute", "transpose", "t", "unsqueeze",
         "squeeze", "empty", "empty_like", "empty_strided", "new_empty", "unbind", "split", "split_with_sizes", "narrow", "_local_scalar_dense",
         "is_same_size", "sym_size", "size", "stride", "lift_fresh", "unfold", "view_as_real", "chunk", "_to_copy_view"}
sites = collections.Counter()
ops = collections.defaultdict(collections.Counter)


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS:
            fr = [f for f in traceback.extract_stack()[:-1] if f.filename.startswith(ROOT) and "launch_sites" not in f.filename]
            where = f"{os.path.relpath(fr[-1].filename, ROOT)}:{fr[-1].lineno} {fr[-1].name}" if fr else "<autograd backward / no repository frame>"
            sites[where] += 1
            ops[where][name] += 1
        return func(*args, **(kwargs or {}))


with Count():
    step()
torch.cuda.synchronize()
print(f"non-view aten ops dispatched in one step: {sum(sites.values())}")
for w, c in sites.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 45):
    print(f"{c:5d}  {w}   " + ", ".join(f"{k} x{n}" for k, n in ops[w].most_common(5)))
