"""Generate tests/golden/collate.npz with the REFERENCE's own DatasetTemplate.collate_batch
(detector3d/pcdet/datasets/dataset.py:175-257) on three ragged samples built by tests/golden/collate_inputs.py.

Run only in the build container (needs /root/reference):  python tests/golden/make_collate_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refimport as R  # noqa: E402

R.import_pcdet()
from pcdet.datasets.dataset import DatasetTemplate  # noqa: E402
from collate_inputs import make_samples  # noqa: E402

ret = DatasetTemplate.collate_batch(make_samples())
out = {k: v for k, v in ret.items() if isinstance(v, np.ndarray)}
out['batch_size'] = np.int64(ret['batch_size'])
np.savez_compressed(os.path.join(HERE, "collate.npz"), **out)
print({k: (v.shape, v.dtype) for k, v in out.items()})
