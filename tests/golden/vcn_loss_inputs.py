"""Deterministic VCN training batch: partial input (B,1024,3), complete cloud (B,4096,3), gt boxes (B,7)."""
import numpy as np


def make_batch(B=3):
    import seevcn_amd.synth as synth
    inp, boxes = synth.make_object_batch(B, seed=1234)
    rng = np.random.default_rng(99)
    complete = np.zeros((B, 4096, 3), np.float32)
    for b in range(B):
        dims = boxes[b, 3:6]
        face = rng.integers(0, 6, 4096)
        uvw = rng.uniform(-0.5, 0.5, (4096, 3))
        uvw[np.arange(4096), face // 2] = np.where(face % 2 == 0, -0.5, 0.5)
        pts = uvw * dims
        c, s = np.cos(boxes[b, 6]), np.sin(boxes[b, 6])
        rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        complete[b] = (pts @ rot.T + boxes[b, :3]).astype(np.float32)
    return inp.astype(np.float32), complete, boxes.astype(np.float32)
