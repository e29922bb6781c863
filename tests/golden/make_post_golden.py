"""Generate tests/golden/vcn_post.npz with the REFERENCE's own get_partial_mesh_batch
(see/surface_completion/models/vcn/utils/sampling.py:8-41,69-81: scipy cKDTree + CPython set) for k = 30 and k = 5.
open3d is not installed, so get_largest_cluster cannot be run: clustering has no golden (parity unpinned, see
oracle/postprocess.py).  Inputs are regenerated from tests/golden/post_inputs.py; only outputs are stored.

Run only in the build container (needs /root/reference):  python tests/golden/make_post_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refimport as R  # noqa: E402

R.import_vcn()
from models.vcn.utils.sampling import get_partial_mesh_batch  # noqa: E402
from post_inputs import make_pairs  # noqa: E402

partial, coarse = make_pairs()
out = {}
for k in (30, 5):
    s = get_partial_mesh_batch(torch.from_numpy(partial), torch.from_numpy(coarse), k=k)
    assert s.dtype == np.float32 and s.shape == partial.shape
    out[f"surface_k{k}"] = s
np.savez_compressed(os.path.join(HERE, "vcn_post.npz"), **out)
print({k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "vcn_post.npz")))
