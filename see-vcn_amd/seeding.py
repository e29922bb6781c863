"""Name-keyed deterministic weights shared by the golden generators, the tests, bench.py and smoke() (no reference import)."""
import numpy as np
import torch


def seeded_state_dict(module, seed=0):
    """Deterministic weights independent of construction order and torch's init code.

    Each tensor is drawn from its own generator seeded by (seed, key) so that the reference
    module here and the build's module on the GPU box get identical values from the *name*.
    Conv/linear weights ~ U(-a, a) with a = sqrt(3/fan_in)*1.2; BN weight ~ U(0.6, 1.4),
    BN running_var ~ U(0.5, 1.5), everything else ~ U(-0.2, 0.2).
    """
    import zlib
    sd = {}
    for k, v in module.state_dict().items():
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) + 1000003 * seed) % (2 ** 31))
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros_like(v)
            continue
        u = torch.rand(v.shape, generator=g, dtype=torch.float32)
        if k.endswith("running_var"):
            t = 0.5 + u
        elif k.endswith("running_mean"):
            t = (u - 0.5) * 0.4
        elif v.dim() >= 2:
            fan_in = int(np.prod(v.shape[1:]))
            a = (3.0 / max(fan_in, 1)) ** 0.5 * 1.2
            t = (u * 2 - 1) * a
        elif k.endswith("weight"):  # 1-D weight = norm scale
            t = 0.6 + 0.8 * u
        else:
            t = (u - 0.5) * 0.4
        sd[k] = t.to(v.dtype)
    return sd
